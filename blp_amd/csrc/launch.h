// launch.h -- host-side launchers implemented in the .hip files, called by the C-ABI in api.cpp.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace blp {

// The filtered setting (train.py:159-171) for a block of Q queries: query q loses the candidates listed in
// val[lo[q] .. hi[q]).  Two layouts share these fields (include/blp_hip.h: blp_filter):
//   CSR       lo = rowptr, hi = rowptr + 1, val = table rows, nothing else set;
//   segments  [lo[q], hi[q]) is a slice of a sorted index of the filtering graph (blp_amd.utils.FilterIndex),
//             val = entity ids, exclude[q] = the triple's own entity (never filtered, utils.py:71,78), ent2idx maps
//             ids to table rows (-1 or beyond its length: not a candidate, utils.py:72,79).
// row_base is subtracted from every row and rows outside [0, N) are skipped (candidate shards).
// The query vectors of a block, q_fixed / q_rel, are (Q, D) row arrays -- either DENSE (row q = base + q * D) or INDEXED
// (row q = base + idx[q] * ld: the fixed-entity vector is a row of the entity table, the relation vector a row of
// rel_emb).  Indexed queries cost no gather and no 2 x Q x D x 4 bytes that every prep / true-key / refinement /
// filter kernel would stream again; their rows come out of L2 instead.  Kernels address both forms through row(q)
// (and flat(offset, D) for the few element-wise sweeps, D the caller's -- usually compile-time -- width).
struct QRows {
    const float* base = nullptr;
    const int64_t* idx = nullptr;  // NULL: dense
    int64_t ld = 0;                // row stride in floats (dense: D)
    __host__ __device__ const float* row(int64_t q) const { return base + (idx ? idx[q] : q) * ld; }
    __host__ __device__ const float* flat(int64_t off, int D) const {
        const int64_t q = off / D;
        return row(q) + (off - q * D);
    }
    static QRows dense(const float* p, int D) { return make(p, nullptr, D); }
    static QRows rows_of(const float* table, const int64_t* idx, int64_t ld) { return make(table, idx, ld); }
    static QRows make(const float* p, const int64_t* idx, int64_t ld) {
        QRows r;
        r.base = p; r.idx = idx; r.ld = ld;
        return r;
    }
};

struct FilterSpec {
    const int64_t* lo = nullptr;
    const int64_t* hi = nullptr;
    const int64_t* val = nullptr;
    const int64_t* exclude = nullptr;
    const int64_t* ent2idx = nullptr;
    int64_t ent2idx_len = 0;
    int64_t row_base = 0;
    __host__ __device__ bool on() const { return lo != nullptr; }
};

bool rank_sad_wide_applicable(int model, int D, int64_t q_head, int64_t q_tail);
size_t rank_all_workspace_bytes(int model, int D, int64_t N, int64_t q_head, int64_t q_tail);

hipError_t launch_rank_all(int model, int D, const float* table, int64_t N, int64_t ld,
                           const QRows q_fixed, const QRows q_rel,
                           const QRows q_true, int64_t q_head, int64_t q_tail,
                           const FilterSpec& filter, int32_t* counts,
                           void* workspace, int n_cu, hipStream_t stream, hipEvent_t ev_start = nullptr,
                           hipEvent_t ev_stop = nullptr);

// rank_all.hip: many passes of <= 4 + 4 queries back to back (one preparation and one finalisation launch for all of them)
bool rank_static_passes_applicable(int model, int D, int64_t N, int64_t batch);
size_t rank_static_passes_workspace_bytes(int D, int64_t n, int64_t batch);
// (table: f32, or a 16-bit table -- dtype: table_elem.h -- when rank_stream16_takes_passes says so)
hipError_t launch_rank_static_passes(int model, int D, const void* table, int dtype, int64_t N, int64_t ld, const QRows q_fixed,
                                     const QRows q_rel, const QRows q_true, int64_t n, int64_t batch, const FilterSpec& filter,
                                     int32_t* counts, void* workspace, int n_cu, hipStream_t stream, hipEvent_t ev_start,
                                     hipEvent_t ev_stop);

// queries.hip: every batch of the reference's evaluation loop in one call (include/blp_hip.h: blp_rank_all_batches)
size_t rank_all_batches_workspace_bytes(int model, int D, int64_t N, int64_t n, int64_t batch, int64_t block_triples);
int64_t rank_all_batches_passes_per_launch(int model, int D, int64_t N, int64_t ld, int64_t n, int64_t batch, int64_t block_triples);
hipError_t launch_rank_all_batches(int model, int D, const float* table, int64_t N, int64_t ld, const float* source, int64_t ld_src,
                                   const int64_t* fixed_row, const float* rel_emb, const int64_t* rel_id, const int64_t* true_row,
                                   int64_t n, int64_t batch, int64_t block_triples, const FilterSpec& filter, int32_t* counts,
                                   void* workspace, int n_cu, hipStream_t stream, hipEvent_t ev_start = nullptr,
                                   hipEvent_t ev_stop = nullptr);

// ... with the candidate table in a 16-bit storage type (include/blp_hip.h: blp_rank_all_batches)
bool rank_all_batches_native16(int model, int D, int64_t N, int64_t ld, int64_t n, int64_t batch, int64_t block_triples);
size_t rank_all_batches16_workspace_bytes(int model, int D, int64_t N, int64_t ld, int64_t n, int64_t batch, int64_t block_triples);
hipError_t launch_rank_all_batches16(int model, int D, const void* table, int dtype, int64_t N, int64_t ld, const float* source,
                                     int64_t ld_src, const int64_t* fixed_row, const float* rel_emb, const int64_t* rel_id,
                                     const int64_t* true_row, int64_t n, int64_t batch, int64_t block_triples, const FilterSpec& filter,
                                     int32_t* counts, void* workspace, int n_cu, hipStream_t stream, hipEvent_t ev_start = nullptr,
                                     hipEvent_t ev_stop = nullptr);

hipError_t launch_rank_metrics(const int32_t* counts, int64_t Q, const int32_t* k, float* rr,
                               uint8_t* hits, hipStream_t stream);

hipError_t launch_rank_from_scores(const float* scores, int64_t Q, int64_t N, int64_t ld, const int64_t* true_idx,
                                   const float* true_score, const int64_t* rowptr, const int64_t* col,
                                   int32_t* counts, hipStream_t stream);

// queries.hip: blp_build_queries (include/blp_hip.h: blp_queries, same fields)
struct QueryBuild {
    const int64_t* triples; int64_t n, block;
    const int64_t* ent2idx; int64_t ent2idx_len;
    const float* source; int64_t src_rows, ld; int D;
    const float* rel_emb; int64_t R;
    const int64_t* heads_key; int64_t n_heads;
    const int64_t* tails_key; int64_t n_tails;
    int64_t index_R;
    float* q_fixed; float* q_rel; int64_t* true_row; int64_t* rel_ids; int* ids_min;
    int64_t* seg_lo; int64_t* seg_hi; int64_t* exclude;
    int64_t* fixed_row;
    int64_t by_position;  // source = [head vectors of the n triples | tail vectors] (launch_gather_triple_vectors)
};
hipError_t launch_build_queries(const QueryBuild& a, hipStream_t stream);
hipError_t launch_gather_triple_vectors(const int64_t* triples, int64_t n, const int64_t* ent2idx, int64_t ent2idx_len,
                                        const void* table, int dtype, int64_t N, int D, int64_t ld, int64_t row_base, float* out,
                                        hipStream_t stream);

bool project_rows_supported(int E, int D);
hipError_t launch_project_rows(const float* x, int64_t n, int64_t ldx, const float* w, int E, int D, int normalize,
                               float* out, int64_t ldo, hipStream_t stream);

bool bow_rows_supported(int E);
hipError_t launch_bow_rows(const int64_t* tok, const float* mask, int64_t n, int L, const float* emb, int64_t V, int E,
                           int normalize, float* out, int64_t ldo, int* bad_tok, hipStream_t stream);

bool dkrl_rows_supported(int E, int D, int L);
hipError_t launch_dkrl_rows(const int64_t* tok, const float* mask, int64_t n, int L, const float* emb, int64_t V, int E,
                            const float* w1, const float* b1, const float* w2, const float* b2, int normalize, float* out,
                            int64_t ldo, int* bad_tok, int n_cu, hipStream_t stream);

struct StridedRows {  // row(i0, i1) = base + i0 * s0 + i1 * s1   (strides in floats)
    const float* base;
    int64_t s0, s1;
};

hipError_t launch_rank_metric_sums(const int32_t* counts, int64_t Q, const int32_t* k, double* sums,
                                   hipStream_t stream);

hipError_t launch_score_fwd(int model, int D, int64_t M0, int64_t M1, StridedRows h, StridedRows t,
                            StridedRows r, float* out, hipStream_t stream);
hipError_t launch_score_bwd(int model, int D, int64_t M0, int64_t M1, StridedRows h, StridedRows t,
                            StridedRows r, const float* grad_out, float* grad_h, float* grad_t,
                            float* grad_r, hipStream_t stream);

hipError_t launch_inbatch_loss_fwd(int model, int loss, int ent_dtype, int rel_dtype, const void* ent, const void* rel,
                                   const int64_t* neg_idx, int B, int K, int D, float regularizer,
                                   float* out_loss, float* save_pos, float* save_neg, unsigned* ticket, hipStream_t stream);
size_t inbatch_loss_save_floats(int model, int B, int K, int D);
int inbatch_loss_fwd_launches(int model, int B, int K, int D, bool regularised);
hipError_t launch_inbatch_loss_bwd(int model, int loss, int ent_dtype, int rel_dtype, const void* ent, const void* rel,
                                   const int64_t* neg_idx, int B, int K, int D, float regularizer,
                                   const float* grad_loss, const float* save_pos, const float* save_neg,
                                   void* grad_ent, void* grad_rel, hipStream_t stream);

}  // namespace blp
