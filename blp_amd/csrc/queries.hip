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

#include "launch.h"

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
    if (!a.q_fixed) return;  // index form only (blp_rank_all_idx): no vectors
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
__global__ __launch_bounds__(256) void gather_triple_vectors_kernel(const int64_t* __restrict__ triples, int64_t n,
                                                                    const int64_t* __restrict__ ent2idx, int64_t ent2idx_len,
                                                                    const float* __restrict__ table, int64_t N, int D, int64_t ld,
                                                                    int64_t row_base, float* __restrict__ out) {
    const int sub = threadIdx.x & 31;
    const int64_t p = (int64_t)blockIdx.x * 8 + (threadIdx.x >> 5);
    if (p >= 2 * n) return;
    const int64_t id = p < n ? triples[3 * p] : triples[3 * (p - n) + 1];
    int64_t row = id;
    if (ent2idx) row = (uint64_t)id < (uint64_t)ent2idx_len ? ent2idx[id] : -1;
    row = row < 0 ? -1 : row - row_base;
    const bool mine = (uint64_t)row < (uint64_t)N;
    const float* src = table + (mine ? row : 0) * ld;
    float* dst = out + p * D;
    for (int c = 4 * sub; c < D; c += 128) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (mine) v = *reinterpret_cast<const float4*>(src + c);
        *reinterpret_cast<float4*>(dst + c) = v;
    }
}

hipError_t launch_gather_triple_vectors(const int64_t* triples, int64_t n, const int64_t* ent2idx, int64_t ent2idx_len,
                                        const float* table, int64_t N, int D, int64_t ld, int64_t row_base, float* out,
                                        hipStream_t stream) {
    if (n == 0) return hipSuccess;
    const int64_t blocks = (2 * n + 7) / 8;
    if (blocks > 0x7fffffff) return hipErrorInvalidValue;
    gather_triple_vectors_kernel<<<dim3((unsigned)blocks), 256, 0, stream>>>(triples, n, ent2idx, ent2idx_len, table, N, D, ld,
                                                                             row_base, out);
    return hipGetLastError();
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
