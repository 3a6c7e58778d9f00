// queries.hip -- the evaluation loop's per-batch prelude (train.py:132-145, utils.py:46-83) for a whole set of triples
// in ONE kernel: what blp_amd.ranking.rank_triples otherwise does with ~35 small torch kernels per evaluation (index,
// gather, cat, searchsorted, reduce: 250 us of an FB15k-237 evaluation whose ranking pass takes 1.05 ms for DistMult).
//
//   heads = ent2idx[triples[:, 0]]; tails = ent2idx[triples[:, 1]]; assert min >= 0       train.py:134-138
//   head queries: fixed = ent_emb[tails], true = heads;  tail queries: fixed = ent_emb[heads], true = tails
//   rel_embs = model.rel_emb(rels)                                                          train.py:141-145
//   the filter of a query = the slice of the sorted (entity, relation) index its key selects  utils.py:46-83
//
// Layout of the 2n queries: block after block of `block` triples, each block as [its head-replacing queries | its
// tail-replacing queries] (train.py:149 order inside a block), so that a block is a contiguous range of every output.
// 32 lanes per query: each moves 16 bytes of the fixed-entity row and of the relation row (D <= 128 floats per
// sweep; wider rows: more sweeps); the lower / upper bound searches of the queries' keys run in workgroups of their own.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "knobs.h"
#include "launch.h"
#include "rank_common.h"
#include "table_elem.h"

namespace blp {

__device__ __forceinline__ int64_t bound(const int64_t* __restrict__ keys, int64_t n, int64_t key, bool upper) {
    int64_t lo = 0, hi = n;  // first index with keys[i] >= key (lower) / > key (upper)
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        const int64_t k = keys[mid];
        if (upper ? k <= key : k < key) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// position p of the layout -> (head side?, triple)
__device__ __forceinline__ void query_at(const QueryBuild& a, int64_t p, bool& head_side, int64_t& t) {
    const int64_t bi = p / (2 * a.block), first = bi * a.block;
    const int64_t nb = a.n - first < a.block ? a.n - first : a.block;  // triples in this block
    const int64_t o = p - 2 * first;
    head_side = o < nb;
    t = first + (head_side ? o : o - nb);
}

// Workgroups [0, gather_blocks): 8 queries each, 32 lanes per query (16 bytes of each row per lane and sweep).
// Workgroups after them: the filter segments, one thread per (query, bound) -- 18 dependent loads per search, so they
// get threads of their own instead of stalling a gathering wave (72 -> 35 us for the FB15k-237 test set).
__global__ __launch_bounds__(256) void build_queries_kernel(QueryBuild a, unsigned gather_blocks) {
    if (blockIdx.x >= gather_blocks) {
        const int64_t i = (int64_t)(blockIdx.x - gather_blocks) * 256 + threadIdx.x;
        const int64_t p = i >> 1;
        if (p >= 2 * a.n) return;
        const bool upper = i & 1;
        bool head_side;
        int64_t t;
        query_at(a, p, head_side, t);
        const int64_t h_id = a.triples[3 * t], t_id = a.triples[3 * t + 1], r_id = a.triples[3 * t + 2];
        // head side: key (tail, rel) in heads_key; tail side: key (head, rel) in tails_key, whose values follow the head
        // side's in the caller's concatenated value array
        const int64_t* keys = head_side ? a.heads_key : a.tails_key;
        const int64_t nk = head_side ? a.n_heads : a.n_tails, base = head_side ? 0 : a.n_heads;
        const bool known = r_id >= 0 && r_id < a.index_R;  // a relation the index never saw matches nothing
        const int64_t key = (head_side ? t_id : h_id) * a.index_R + r_id;
        const int64_t b = known ? bound(keys, nk, key, upper) : 0;
        if (upper) {
            a.seg_hi[p] = base + b;
        } else {
            a.seg_lo[p] = base + b;
            a.exclude[p] = head_side ? h_id : t_id;  // the triple's own entity is never filtered (utils.py:71,78)
        }
        return;
    }
    const int sub = threadIdx.x & 31;
    const int64_t p = (int64_t)blockIdx.x * 8 + (threadIdx.x >> 5);  // query position
    if (p >= 2 * a.n) return;
    bool head_side;
    int64_t t;
    query_at(a, p, head_side, t);
    const int64_t h_id = a.triples[3 * t], t_id = a.triples[3 * t + 1], r_id = a.triples[3 * t + 2];
    auto row_of = [&](int64_t id) -> int64_t {
        if (!a.ent2idx) return (uint64_t)id < (uint64_t)a.src_rows ? id : -1;
        const int64_t r = (uint64_t)id < (uint64_t)a.ent2idx_len ? a.ent2idx[id] : -1;
        return (uint64_t)r < (uint64_t)a.src_rows ? r : -1;
    };
    const int64_t h_row = row_of(h_id), t_row = row_of(t_id);
    int64_t fixed_row = head_side ? t_row : h_row, true_row = head_side ? h_row : t_row;
    if (a.by_position) {  // `source` = [head vectors of the n triples | tail vectors]: positions instead of table rows
        fixed_row = fixed_row < 0 ? -1 : (head_side ? a.n + t : t);
        true_row = true_row < 0 ? -1 : (head_side ? t : a.n + t);
    }
    const bool rel_ok = (uint64_t)r_id < (uint64_t)a.R;
    if (sub == 0) {  // (a bad id: row / relation 0, so that nothing downstream reads out of bounds; ids_min tells)
        a.true_row[p] = true_row >= 0 ? true_row : 0;
        a.rel_ids[p] = rel_ok ? r_id : 0;
        if (a.fixed_row) a.fixed_row[p] = fixed_row >= 0 ? fixed_row : 0;
        if (h_row < 0 || t_row < 0 || !rel_ok) atomicMin(a.ids_min, -1);  // train.py:137-138's assertion, left on the device
    }
    if (!a.q_fixed) return;  // index form only (blp_rank_all_shard / blp_rank_all_batches): no vectors
    // the two vectors: fixed entity and relation (an id that is no row: zeros -- the caller checks ids_min)
    const float* fsrc = a.source + (fixed_row >= 0 ? fixed_row : 0) * a.ld;
    const float* rsrc = a.rel_emb + (rel_ok ? r_id : 0) * (int64_t)a.D;
    float* fdst = a.q_fixed + p * a.D;
    float* rdst = a.q_rel + p * a.D;
    for (int c = 4 * sub; c < a.D; c += 128) {
        float4 f = make_float4(0.f, 0.f, 0.f, 0.f), r = f;
        if (fixed_row >= 0) f = *reinterpret_cast<const float4*>(fsrc + c);
        if (rel_ok) r = *reinterpret_cast<const float4*>(rsrc + c);
        *reinterpret_cast<float4*>(fdst + c) = f;
        *reinterpret_cast<float4*>(rdst + c) = r;
    }
}

// blp_gather_triple_vectors: out[t] = the head's vector of triple t, out[n + t] = its tail's, for the rows this shard owns
// (global row - row_base in [0, N)); zeros otherwise, so that one all-reduce over the ranks replicates every vector.
// 32 lanes per vector, 16 bytes per lane and sweep.
template <class TE>  // TE: the table's storage type (table_elem.h); the vectors come out in f32 (widened exactly)
__global__ __launch_bounds__(256) void gather_triple_vectors_kernel(const int64_t* __restrict__ triples, int64_t n,
                                                                    const int64_t* __restrict__ ent2idx, int64_t ent2idx_len,
                                                                    const TE* __restrict__ table, int64_t N, int D, int64_t ld,
                                                                    int64_t row_base, float* __restrict__ out) {
    const int sub = threadIdx.x & 31;
    const int64_t p = (int64_t)blockIdx.x * 8 + (threadIdx.x >> 5);
    if (p >= 2 * n) return;
    const int64_t id = p < n ? triples[3 * p] : triples[3 * (p - n) + 1];
    int64_t row = id;
    if (ent2idx) row = (uint64_t)id < (uint64_t)ent2idx_len ? ent2idx[id] : -1;
    row = row < 0 ? -1 : row - row_base;
    const bool mine = (uint64_t)row < (uint64_t)N;
    const TE* src = table + (mine ? row : 0) * ld;
    float* dst = out + p * D;
    for (int c = 4 * sub; c < D; c += 128) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (mine) v = load4<TE>(src + c);
        *reinterpret_cast<float4*>(dst + c) = v;
    }
}

hipError_t launch_gather_triple_vectors(const int64_t* triples, int64_t n, const int64_t* ent2idx, int64_t ent2idx_len,
                                        const void* table, int dtype, int64_t N, int D, int64_t ld, int64_t row_base, float* out,
                                        hipStream_t stream) {
    if (n == 0) return hipSuccess;
    const int64_t blocks = (2 * n + 7) / 8;
    if (blocks > 0x7fffffff) return hipErrorInvalidValue;
    const dim3 grid((unsigned)blocks);
    if (dtype == kTableF32)
        gather_triple_vectors_kernel<float><<<grid, 256, 0, stream>>>(triples, n, ent2idx, ent2idx_len, static_cast<const float*>(table), N,
                                                                      D, ld, row_base, out);
    else if (dtype == kTableF16)
        gather_triple_vectors_kernel<_Float16><<<grid, 256, 0, stream>>>(triples, n, ent2idx, ent2idx_len,
                                                                         static_cast<const _Float16*>(table), N, D, ld, row_base, out);
    else if (dtype == kTableBF16)
        gather_triple_vectors_kernel<__bf16><<<grid, 256, 0, stream>>>(triples, n, ent2idx, ent2idx_len, static_cast<const __bf16*>(table),
                                                                       N, D, ld, row_base, out);
    else
        return hipErrorInvalidValue;
    return hipGetLastError();
}

// A 16-bit table widened into a packed (N, D) f32 copy: what every route but the few-queries ring kernels ranks a 16-bit table
// through (they are bound by arithmetic, not by the table read; their operand images are made from f32 rows).
template <class TE>
__global__ __launch_bounds__(256) void widen_table_kernel(const TE* __restrict__ table, int64_t N, int D, int64_t ld,
                                                          float* __restrict__ out) {
    const int64_t quads = N * (D / 4), stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < quads; i += stride) {
        const int64_t r = i / (D / 4), c = (i - r * (D / 4)) * 4;
        *reinterpret_cast<float4*>(out + r * D + c) = load4<TE>(table + r * ld + c);
    }
}
static hipError_t launch_widen_table(const void* table, int dtype, int64_t N, int D, int64_t ld, float* out, hipStream_t stream) {
    if (N == 0) return hipSuccess;
    const int64_t want = (N * (D / 4) + 255) / 256;
    const unsigned blocks = (unsigned)(want < 16384 ? want : 16384);
    if (dtype == kTableF16)
        widen_table_kernel<_Float16><<<blocks, 256, 0, stream>>>(static_cast<const _Float16*>(table), N, D, ld, out);
    else if (dtype == kTableBF16)
        widen_table_kernel<__bf16><<<blocks, 256, 0, stream>>>(static_cast<const __bf16*>(table), N, D, ld, out);
    else
        return hipErrorInvalidValue;
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------------
// blp_rank_all_batches: the reference's loop hands the ranking one batch of eval_batch_size triples at a time
// (train.py:128-157); a query's counts do not depend on the batch it came in, so all batches are ranked as ONE block per
// kBatchesSuper triples: the queries -- given in the loop's own layout, batch after batch, each as [its head-replacing
// queries | its tail-replacing queries] -- are permuted into [all heads | all tails], ranked, and the counts scattered
// back into the loop's layout.  Two tiny kernels around the ranking instead of 827 x (3-5 launches) for FB15k-237's test set.
struct BatchPerm {
    int64_t t0, m, n, batch;  // triples [t0, t0 + m) of n, `batch` per batch
    __device__ __forceinline__ int64_t blocked(int64_t j) const {  // permuted position j -> position in the loop's layout
        const bool tail = j >= m;
        const int64_t t = t0 + (tail ? j - m : j);
        const int64_t first = t / batch * batch, nb = n - first < batch ? n - first : batch;
        return 2 * first + (tail ? nb : 0) + (t - first);
    }
};

__global__ __launch_bounds__(256) void permute_batches_kernel(BatchPerm pm, const int64_t* __restrict__ fixed_row,
                                                             const int64_t* __restrict__ rel_id, const int64_t* __restrict__ true_row,
                                                             const int64_t* __restrict__ seg_lo, const int64_t* __restrict__ seg_hi,
                                                             const int64_t* __restrict__ exclude, int64_t* __restrict__ out) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x, Q = 2 * pm.m;
    if (j >= Q) return;
    const int64_t p = pm.blocked(j);
    out[j] = fixed_row[p];
    out[Q + j] = rel_id[p];
    out[2 * Q + j] = true_row[p];
    if (seg_lo) {
        out[3 * Q + j] = seg_lo[p];
        out[4 * Q + j] = seg_hi[p];
        if (exclude) out[5 * Q + j] = exclude[p];
    }
}

__global__ __launch_bounds__(256) void unpermute_counts_kernel(BatchPerm pm, const int4* __restrict__ permuted, int4* __restrict__ counts) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= 2 * pm.m) return;
    counts[pm.blocked(j)] = permuted[j];
}

constexpr int64_t kBatchesSuper = 65536;  // triples ranked as one block by default (the block size of blp_amd.ranking)
// triples per ranking pass: whole batches, up to block_triples (0: kBatchesSuper) -- at least one batch
static int64_t batches_super(int64_t n, int64_t batch, int64_t block_triples) {
    if (n <= batch) return n;
    const int64_t want = block_triples > 0 ? block_triples : kBatchesSuper;
    const int64_t s = want / batch * batch;
    return s >= batch ? s : batch;
}

// Workspace of the ranking passes of a call that walks n triples `step` at a time.  rank_all_workspace_bytes is NOT monotone
// in the query count -- the route changes with Q x N (small slots, fixed-point / MFMA pre-pass, exact tiles) -- so the short
// last pass can need MORE than a full one: the call's passes share the larger of the two.
static size_t batches_inner_bytes(int model, int D, int64_t N, int64_t n, int64_t step) {
    const int64_t m_full = step < n ? step : n, m_tail = n > step ? n % step : 0;
    size_t inner = rank_all_workspace_bytes(model, D, N, m_full, m_full);
    if (m_tail > 0) {
        const size_t tail = rank_all_workspace_bytes(model, D, N, m_tail, m_tail);
        if (tail > inner) inner = tail;
    }
    return inner;
}

size_t rank_all_batches_workspace_bytes(int model, int D, int64_t N, int64_t n, int64_t batch, int64_t block_triples) {
    if (n <= 0 || batch <= 0) return 0;
    const int64_t super = batches_super(n, batch, block_triples), m = super < n ? super : n;
    const size_t inner = batches_inner_bytes(model, D, N, n, super <= batch ? batch : super);
    if (super <= batch) {  // a pass per batch: every batch is [heads | tails] already
        if (n > batch && rank_static_passes_applicable(model, D, N, batch)) {
            const size_t all = rank_static_passes_workspace_bytes(D, n, batch);
            return all > inner ? all : inner;
        }
        return inner;
    }
    return (inner + 255) / 256 * 256 + (size_t)(6 * 2 * m) * 8 + (size_t)(2 * m) * 16;
}

// A 16-BIT TABLE (table_elem.h).  The reference-batched passes of <= 4 + 4 queries over a long table read it as it is
// (rank_stream16.hip); any other shape of call ranks a widened f32 copy kept at the end of the workspace.
bool rank_all_batches_native16(int model, int D, int64_t N, int64_t ld, int64_t n, int64_t batch, int64_t block_triples) {
    if (n <= 0 || batch <= 0) return false;
    return batches_super(n, batch, block_triples) <= batch && rank_static_passes_applicable(model, D, N, batch) &&
           knob(KNOB_STREAM_KERNEL) != 2 && rank_stream16_takes_passes(model, D, N, ld, batch, n);
}
static size_t batches_front_bytes(int model, int D, int64_t N, int64_t n, int64_t batch, int64_t block_triples) {
    const size_t b = n <= batch ? rank_all_workspace_bytes(model, D, N, n, n) : rank_all_batches_workspace_bytes(model, D, N, n, batch, block_triples);
    return (b + 255) / 256 * 256;
}
size_t rank_all_batches16_workspace_bytes(int model, int D, int64_t N, int64_t ld, int64_t n, int64_t batch, int64_t block_triples) {
    if (n <= 0 || batch <= 0) return 0;
    if (rank_all_batches_native16(model, D, N, ld, n, batch, block_triples)) return rank_static_passes_workspace_bytes(D, n, batch);
    return batches_front_bytes(model, D, N, n, batch, block_triples) + (size_t)N * D * 4;
}
hipError_t launch_rank_all_batches16(int model, int D, const void* table, int dtype, int64_t N, int64_t ld, const float* source,
                                     int64_t ld_src, const int64_t* fixed_row, const float* rel_emb, const int64_t* rel_id,
                                     const int64_t* true_row, int64_t n, int64_t batch, int64_t block_triples, const FilterSpec& filter,
                                     int32_t* counts, void* workspace, int n_cu, hipStream_t stream, hipEvent_t ev_start,
                                     hipEvent_t ev_stop) {
    if (n <= 0) return hipSuccess;
    if (rank_all_batches_native16(model, D, N, ld, n, batch, block_triples))
        return launch_rank_static_passes(model, D, table, dtype, N, ld, QRows::rows_of(source, fixed_row, ld_src),
                                         QRows::rows_of(rel_emb, rel_id, D), QRows::rows_of(source, true_row, ld_src), n, batch, filter,
                                         counts, workspace, n_cu, stream, ev_start, ev_stop);
    float* wide = reinterpret_cast<float*>(static_cast<char*>(workspace) + batches_front_bytes(model, D, N, n, batch, block_triples));
    if (const hipError_t e = launch_widen_table(table, dtype, N, D, ld, wide, stream); e != hipSuccess) return e;
    if (n <= batch)  // one batch: one block of [heads | tails]
        return launch_rank_all(model, D, wide, N, D, QRows::rows_of(source, fixed_row, ld_src), QRows::rows_of(rel_emb, rel_id, D),
                               QRows::rows_of(source, true_row, ld_src), n, n, filter, counts, workspace, n_cu, stream, ev_start, ev_stop);
    return launch_rank_all_batches(model, D, wide, N, D, source, ld_src, fixed_row, rel_emb, rel_id, true_row, n, batch, block_triples,
                                   filter, counts, workspace, n_cu, stream, ev_start, ev_stop);
}

// How many of the call's ranking passes the FIRST ranking launch covers (what blp_profile_next_rank_kernel's events bracket):
// all of them when a ring kernel takes the passes of a reference-batched call in one launch, else one.
int64_t rank_all_batches_passes_per_launch(int model, int D, int64_t N, int64_t ld, int64_t n, int64_t batch, int64_t block_triples) {
    if (n <= 0 || batch <= 0) return 0;
    const int64_t super = batches_super(n, batch, block_triples);
    if (super <= batch && n > batch && rank_static_passes_applicable(model, D, N, batch) &&
        knob(KNOB_STREAM_KERNEL) != 2 && rank_stream_takes_passes(model, D, N, ld, batch, n))
        return (n + batch - 1) / batch;
    return 1;
}

hipError_t launch_rank_all_batches(int model, int D, const float* table, int64_t N, int64_t ld, const float* source, int64_t ld_src,
                                   const int64_t* fixed_row, const float* rel_emb, const int64_t* rel_id, const int64_t* true_row,
                                   int64_t n, int64_t batch, int64_t block_triples, const FilterSpec& filter, int32_t* counts,
                                   void* workspace, int n_cu, hipStream_t stream, hipEvent_t ev_start, hipEvent_t ev_stop) {
    // (ev_start / ev_stop: blp_profile_next_rank_kernel -- they bracket the FIRST ranking pass of the call)
    if (n <= 0) return hipSuccess;
    const int64_t super = batches_super(n, batch, block_triples);
    if (super <= batch) {  // one ranking pass per batch, as the reference's loop runs them: slices of the loop's layout
        if (n > batch && rank_static_passes_applicable(model, D, N, batch))  // <= 4 + 4 queries per pass: one prep, one finalize for all
            return launch_rank_static_passes(model, D, table, kTableF32, N, ld, QRows::rows_of(source, fixed_row, ld_src),
                                             QRows::rows_of(rel_emb, rel_id, D), QRows::rows_of(source, true_row, ld_src), n, batch,
                                             filter, counts, workspace, n_cu, stream, ev_start, ev_stop);
        for (int64_t t0 = 0; t0 < n; t0 += batch) {
            const int64_t m = n - t0 < batch ? n - t0 : batch, q0 = 2 * t0;
            FilterSpec f = filter;
            if (filter.on()) {
                f.lo = filter.lo + q0;
                f.hi = filter.hi + q0;
                f.exclude = filter.exclude ? filter.exclude + q0 : nullptr;
            }
            const hipError_t err = launch_rank_all(model, D, table, N, ld, QRows::rows_of(source, fixed_row + q0, ld_src),
                                                   QRows::rows_of(rel_emb, rel_id + q0, D),
                                                   QRows::rows_of(source, true_row + q0, ld_src), m, m, f, counts + 4 * q0, workspace,
                                                   n_cu, stream, t0 == 0 ? ev_start : nullptr, t0 == 0 ? ev_stop : nullptr);
            if (err != hipSuccess) return err;
        }
        return hipSuccess;
    }
    const int64_t m_max = super < n ? super : n;
    if ((2 * m_max + 255) / 256 > 0x7fffffff) return hipErrorInvalidValue;  // the permutation kernels' grid
    const size_t inner = (batches_inner_bytes(model, D, N, n, super) + 255) / 256 * 256;
    int64_t* perm = reinterpret_cast<int64_t*>(static_cast<char*>(workspace) + inner);
    int4* pcounts = reinterpret_cast<int4*>(perm + 6 * 2 * m_max);
    for (int64_t t0 = 0; t0 < n; t0 += super) {
        const int64_t m = n - t0 < super ? n - t0 : super, Q = 2 * m;
        const BatchPerm pm{t0, m, n, batch};
        const unsigned blocks = (unsigned)((Q + 255) / 256);
        permute_batches_kernel<<<blocks, 256, 0, stream>>>(pm, fixed_row, rel_id, true_row, filter.lo, filter.hi, filter.exclude, perm);
        if (const hipError_t e = hipGetLastError(); e != hipSuccess) return e;
        FilterSpec f = filter;
        if (filter.on()) {
            f.lo = perm + 3 * Q;
            f.hi = perm + 4 * Q;
            f.exclude = filter.exclude ? perm + 5 * Q : nullptr;
        }
        const hipError_t err = launch_rank_all(model, D, table, N, ld, QRows::rows_of(source, perm, ld_src),
                                               QRows::rows_of(rel_emb, perm + Q, D), QRows::rows_of(source, perm + 2 * Q, ld_src),
                                               m, m, f, reinterpret_cast<int32_t*>(pcounts), workspace, n_cu, stream,
                                               t0 == 0 ? ev_start : nullptr, t0 == 0 ? ev_stop : nullptr);
        if (err != hipSuccess) return err;
        unpermute_counts_kernel<<<blocks, 256, 0, stream>>>(pm, pcounts, reinterpret_cast<int4*>(counts));
        if (const hipError_t e = hipGetLastError(); e != hipSuccess) return e;
    }
    return hipSuccess;
}

hipError_t launch_build_queries(const QueryBuild& a, hipStream_t stream) {
    hipError_t err = hipMemsetAsync(a.ids_min, 0, sizeof(int), stream);
    if (err != hipSuccess || a.n == 0) return err;
    const int64_t gather_blocks = (2 * a.n + 7) / 8, search_blocks = a.seg_lo ? (4 * a.n + 255) / 256 : 0;
    if (gather_blocks + search_blocks > 0x7fffffff) return hipErrorInvalidValue;
    build_queries_kernel<<<dim3((unsigned)(gather_blocks + search_blocks)), 256, 0, stream>>>(a, (unsigned)gather_blocks);
    return hipGetLastError();
}

}  // namespace blp
