/*
 * blp_hip.h -- C-ABI of libblp_hip.so: MI355X (gfx950) kernels for the link-prediction
 * scoring / ranking hot path of dfdazac/blp.
 *
 * The reference is Python/PyTorch and has no FFI of its own; this is the boundary a maintainer
 * binds with ctypes (INTEGRATION.md shows the stub).  Each entry point names the reference
 * expression it replaces (file:line under the reference checkout).
 *
 * Conventions (every call):
 *   - plain C types; every pointer except `workspace`-free host outputs is a DEVICE pointer owned
 *     by the caller (torch: tensor.data_ptr()); nothing is allocated or freed inside a call;
 *   - `device` is the HIP device ordinal the pointers live on, `stream` a hipStream_t (torch:
 *     torch.cuda.current_stream().cuda_stream; NULL = the legacy default stream).  Calls only
 *     enqueue work on `stream`: no host synchronisation, safe to capture in a hipGraph (one exception, once per device:
 *     blp_selftest below);
 *   - re-entrant: no process-wide mutable state except the per-device verdict of the matrix-pipe self-test (blp_selftest:
 *     written once, under a mutex; the test knobs exist only in the -DBLP_TEST_HOOKS build, see the end of this file); one thread per device (nn.DataParallel replicas) may call concurrently.  ctypes releases
 *     the GIL for the duration of the call;
 *   - return 0 (BLP_OK) or a negative blp_status; blp_last_error() gives the thread-local message;
 *   - f32 tensors, int64 indices exactly as the reference produces them (neg_idx, true_idx);
 *   - arithmetic follows the torch-CPU evaluation order of the reference expressions (see
 *     oracle/blp_oracle.c), so scores are bit-identical and rank counts are exact.
 */
#ifndef BLP_HIP_H
#define BLP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BLP_HIP_VERSION 60000 /* major*10000 + minor*100 + patch; 6.0.0: the in-batch loss in two launches (fwd takes a ticket
                                 counter, save_pos sized by blp_inbatch_loss_save_floats); entry points pruned: ONE form per
                                 operation (storage-typed, filter as blp_filter), q_rel_id gone */

typedef enum blp_status {
    BLP_OK = 0,
    BLP_ERR_BAD_ARG = -1,         /* null / misaligned pointer, negative size, bad enum     */
    BLP_ERR_UNSUPPORTED_DIM = -2, /* D not in the compiled set (see blp_dim_supported)      */
    BLP_ERR_HIP = -3,             /* a HIP runtime call failed; message has hipGetErrorString */
    BLP_ERR_WORKSPACE = -4        /* workspace NULL / too small / misaligned                */
} blp_status;

/* rel_model of models.py:16-26 */
typedef enum blp_model { BLP_TRANSE = 0, BLP_DISTMULT = 1, BLP_COMPLEX = 2, BLP_SIMPLE = 3 } blp_model;
/* loss_fn of models.py:31-36 */
typedef enum blp_loss { BLP_LOSS_MARGIN = 0, BLP_LOSS_NLL = 1 } blp_loss;

typedef struct blp_caps {
    int compute_units;        /* 256 on MI355X                      */
    int wavefront_size;       /* 64                                 */
    int lds_bytes_per_cu;     /* 163840                             */
    int clock_mhz;            /* max engine clock                   */
    int64_t hbm_bytes;        /* total device memory                */
    char arch[32];            /* "gfx950..."                        */
    int mfma_bf16_accum;      /* verdict of the matrix-pipe self-test on this device (blp_selftest): 0 = not run yet,
                                 1 = passed (bilinear blocks use the bf16 x 3 MFMA pre-pass), 2 = FAILED (they use the
                                 f32-chain pre-pass, ~3.5 x slower, same results)                              */
    float mfma_bf16_accum_worst; /* the largest |S~ - S3| / (262 u T) the self-test measured; it fails at 0.5 */
} blp_caps;

int blp_version(void);
const char *blp_last_error(void);
int blp_device_caps(int device, blp_caps *out);

/* The matrix-pipe self-test behind the bilinear pre-pass (DistMult / ComplEx / SimplE blocks of >= 32 queries).  That
 * pre-pass decides most (query, candidate) pairs from a bf16 x 3 MFMA product inside an error band; every term of the
 * band is arithmetic except one: how v_mfma_f32_32x32x16_bf16 rounds its 16-term accumulation, which the ISA does not
 * document.  Once per device this library therefore pushes adversarial operand sets (same-sign runs, one huge term among
 * tiny ones, exact cancellation, 2^-60 / 2^50 scales, all-ones significands) through the pre-pass's own MFMA sequence
 * and compares every accumulator with the exactly evaluated split sum against HALF of what the band allows.  A device
 * that fails is served by the f32-chain pre-pass (exact fused multiply-adds; its band needs no assumption) -- the results
 * are the same either way, only the speed differs.
 *   blp_selftest(device, stream)  runs it now if it has not run on `device` (allocates and frees 4 096 bytes, enqueues on
 *        `stream` and WAITS for it -- a set-up call, not to be captured into a graph); returns the verdict (1 / 2) or a
 *        negative status.  CALL IT AT SET-UP (blp_amd.ops does, before the first bilinear ranking call of a device): a
 *        caller that does not leaves it to the first bilinear blp_rank_all* call on the device, which then runs it by
 *        itself with its workspace as scratch -- that ONE call holds a process-wide mutex and synchronises `stream`
 *        (several threads making their first calls together are serialised by it); if `stream` is being captured the
 *        verdict stays open and that call takes the f32-chain pre-pass.
 * This per-device verdict (an int, written once under a mutex) is the library's only process-wide state. */
int blp_selftest(int device, void *stream);
/* storage types of the arrays that may come in 16 bits (the in-batch loss' embeddings; the candidate table of
 * blp_rank_all_batches) */
enum { BLP_DTYPE_F32 = 0, BLP_DTYPE_F16 = 1, BLP_DTYPE_BF16 = 2 };

/* 1 if kernels are compiled for embedding width D of `model` (D % 32 == 0, D <= 256; the
 * half-width models complex / simple additionally need D % 64 == 0). */
int blp_dim_supported(int model, int D);

/* --------------------------------------------------------------------------------------------
 * All-entities ranking  (replaces train.py:146-171 + utils.py:103-105 for one block of queries)
 *
 *   heads_predictions = score_fn(ent_emb, tail_embs, rel_embs)      train.py:146
 *   tails_predictions = score_fn(head_embs, ent_emb, rel_embs)      train.py:147
 *   pred_ents = cat(heads_predictions, tails_predictions)           train.py:149
 *   best = (pred > true).sum + 1 ; worst = (pred >= true).sum       utils.py:103-105
 *   pred_ents[filter_mask] = pred_ents.min() - 1.0 ; get_metrics    train.py:165-167
 *
 * One call streams the (N, D) candidate table once and serves Q = q_head + q_tail queries:
 * queries [0, q_head) replace the HEAD (q_fixed = tail embedding), queries [q_head, Q) replace
 * the TAIL (q_fixed = head embedding) -- the same "head queries first" order as train.py:149.
 * The (Q, N) score matrix is never materialised.  Results are the reference's, bit for bit, whichever
 * kernel serves the block: exact f32 kernels for small blocks; for blocks of many
 * queries a cheap pre-pass (16-bit fixed-point v_sad_u16 for TransE, bf16 x 3 MFMA GEMM for the
 * bilinear models) with a rigorous error band, then exact re-scoring of the undecided pairs -- and when a pre-pass leaves
 * more undecided than its lists hold (exact ties with the true entity on whole percents of the table), a device-side counter
 * makes the exact kernels re-rank the block instead: a block never costs more than pre-pass + exact kernel.
 *
 *   table      (N, D) f32, row stride ld floats (ld % 4 == 0, 16-byte aligned base)
 *   q_fixed    (Q, D) f32  the entity kept fixed (tail_embs for head queries, head_embs for tail)
 *   q_rel      (Q, D) f32  rel_emb(rels)
 *   true_row   (Q) int64   row of the true entity in `table` (true_ents, train.py:150), or NULL
 *   q_true     (Q, D) f32  the true entity's vector, used when true_row == NULL (candidate-axis
 *                          sharding: the true row may live in another shard).  Exactly one of
 *                          true_row / q_true is non-NULL.
 *   filter     a blp_filter (below): the table rows the filtered setting removes for each query (the True entries of
 *                          utils.get_triple_filters' mask, utils.py:46-83) as segments of a sorted index of the filtering
 *                          graph -- or as a plain CSR (seg_lo = rowptr, seg_hi = rowptr + 1, values = col, the rest
 *                          NULL / 0); NULL = no filtering (filtered == raw).  Each row at most once per query (a mask
 *                          bit is set once however many parallel edges the graph has), never the true entity
 *                          (utils.py:71,78); blp_amd.utils.FilterIndex produces exactly that.  Rows outside [0, N) are
 *                          ignored, so a candidate shard holding global rows [lo, lo + N) takes the global filter
 *                          with row_base = lo.
 *   counts     (Q, 4) int32 OUT: {#(pred > true), #(pred >= true), same two over the
 *                          non-filtered candidates}.  Overwritten.  With the candidate axis
 *                          sharded, per-shard counts add up to the unsharded ones.
 *   workspace  caller-owned scratch of >= blp_rank_all_workspace_bytes(model, N, D, q_head, q_tail)
 *              bytes, 256-B aligned.
 * Limits: Q <= 2^30, N < 2^31 per call (int32 counts); q_fixed / q_rel / q_true 16-byte aligned.
 * -------------------------------------------------------------------------------------------- */
size_t blp_rank_all_workspace_bytes(int model, int64_t N, int D, int64_t q_head, int64_t q_tail);

/* 1 if blp_rank_all takes a block of this shape: D in {64, 128, 256} (what blp_dim_supported reports), or
 * TransE at any D % 4 == 0, D <= 1024 (the 300 / 768-wide bag-of-words and DKRL encoders, models.py:118-135,
 * 165-172).  Otherwise callers use blp_score_fwd + blp_rank_from_scores. */
int blp_rank_all_supported(int model, int D, int64_t q_head, int64_t q_tail);
/* The filter of a ranking call: SEGMENTS of a sorted index of the filtering graph, so that
 * utils.get_triple_filters (utils.py:46-83: a Python walk over a networkx graph per batch, then a dense
 * (2B, N) mask copied to the device) needs no per-batch list at all:
 *   the graph's edges are sorted once by key (head, rel) with value tail -- and by (tail, rel) with value
 *   head -- into one `values` array of ENTITY IDS (blp_amd.utils.FilterIndex); for query q the caller
 *   binary-searches the key of its triple and passes the slice [seg_lo[q], seg_hi[q]) of `values`;
 *   exclude[q] is the triple's own entity at the replaced position, which is never filtered (utils.py:71,78);
 *   ent2idx (utils.make_ent2idx, utils.py:31-43) maps an entity id to its table row, -1 (or an id beyond
 *   ent2idx_len) = not a candidate (utils.py:72,79); NULL = the values are table rows already;
 *   row_base is subtracted from every row and rows outside [0, N) are skipped (a candidate shard).
 * A value may occur once per segment (parallel edges collapsed when the index is built).
 * A CSR (rowptr (Q + 1), col (nnz) of table rows) is the special case seg_lo = rowptr, seg_hi = rowptr + 1, values = col,
 * exclude = ent2idx = NULL, row_base = 0.
 * filter == NULL: no filtering. */
typedef struct blp_filter {
    const int64_t *seg_lo;   /* (Q) */
    const int64_t *seg_hi;   /* (Q) */
    const int64_t *values;
    const int64_t *exclude;  /* (Q) or NULL */
    const int64_t *ent2idx;  /* (ent2idx_len) or NULL */
    int64_t ent2idx_len;
    int64_t row_base;
} blp_filter;
int blp_rank_all(int model, const float *table, int64_t N, int D, int64_t ld,
                 const float *q_fixed, const float *q_rel, const int64_t *true_row, const float *q_true,
                 int64_t q_head, int64_t q_tail, const blp_filter *filter, int32_t *counts,
                 void *workspace, size_t workspace_bytes, int device, void *stream);

/* The same with the queries given as INDICES instead of vectors, for ONE SHARD OF THE CANDIDATE AXIS (the north_star's
 * multi-GPU layout: the entity table partitioned by rows over the ranks, every rank ranking every query against its own
 * rows) -- or, with source == table, for the whole table: query q's fixed-entity vector is row fixed_row[q] of `source`, its
 * relation vector row rel_id[q] of rel_emb (R, D) f32 contiguous -- exactly what the reference gathers into ent_emb[tails] /
 * rel_emb(rels) (train.py:141-145), un-gathered: no (Q, D) arrays are built or streamed.  `table` is this rank's shard -- global rows
 * [row_base, row_base + N), the CANDIDATES -- while the queries' own vectors, the fixed entity and the true entity of every
 * triple, are rows of a second array `source` (S, D), row stride ld_src, that every rank holds in full: either the whole
 * table (small tables: one all-gather per evaluation) or the vectors of the entities that occur in the triples
 * (blp_gather_triple_vectors + one all-reduce).  fixed_row / true_row (Q,) index `source`; nothing else changes: the
 * reference's `ent_emb[tails]`, `ent_emb[heads]`, `rel_emb(rels)` (train.py:141-145) stay un-gathered, `true_ents`
 * (train.py:150) becomes the true entity's score from its vector, and the filter's row_base keeps the rows of other shards
 * out (blp_filter).  Per-shard counts add up exactly to the unsharded counts: one all-gather of (Q, 4) int32 + a sum is
 * the only exchange after the ranking.  The unsharded case: source == table, S == N, ld_src == ld.
 * Workspace: blp_rank_all_workspace_bytes(model, N, D, q_head, q_tail). */
int blp_rank_all_shard(int model, const float *table, int64_t N, int D, int64_t ld, const float *source, int64_t S,
                       int64_t ld_src, const int64_t *fixed_row, const float *rel_emb, int64_t R, const int64_t *rel_id,
                       const int64_t *true_row, int64_t q_head, int64_t q_tail, const blp_filter *filter, int32_t *counts,
                       void *workspace, size_t workspace_bytes, int device, void *stream);

/* EVERY BATCH OF THE REFERENCE'S EVALUATION LOOP IN ONE CALL.  train.py:128-157 hands the ranking eval_batch_size triples at
 * a time (64 in the FB15k-237 scripts, 2 for Wikidata5M: scripts/blp-*.sh:18); a query's counts do not depend on the batch
 * it came in, so a caller that keeps the loop's layout passes all of it at once: the 2 n_triples queries batch after batch
 * of `batch` triples (the last may be short), each batch as [its head-replacing queries | its tail-replacing queries] --
 * exactly what blp_build_queries writes with block = batch.  fixed_row / rel_id / true_row / the filter's seg_lo, seg_hi,
 * exclude and `counts` (2 n_triples, 4) all use that layout; source / S / ld_src as in blp_rank_all_shard (source = table
 * for an unsharded table).  Inside, the queries are ranked as ONE block per block_triples triples (0 = the default, 65 536;
 * whole batches; [all heads | all tails]: the throughput-bound kernels) and the counts scattered back: 3.1 ms instead of
 * 827 calls x 29 us for the FB15k-237 test set.  block_triples <= batch keeps one ranking pass per batch -- the reference's
 * own pass structure (one read of the table per eval_batch_size triples), issued back to back without a host round trip.
 *
 * THE CANDIDATE TABLE MAY COME IN A 16-BIT STORAGE TYPE.  The table build can emit a half-precision copy of the entity table
 * next to the f32 one (train.py:96-121 builds `ent_emb` once per evaluation); `table` (N, D) is f32 (BLP_DTYPE_F32, row
 * stride ld floats, ld % 4 == 0), IEEE half (BLP_DTYPE_F16) or bfloat16 (BLP_DTYPE_BF16; row stride ld ELEMENTS, ld % 8 ==
 * 0), 16-byte aligned.  A 16-bit element widens to f32 exactly: every kernel widens first and then runs the f32 arithmetic
 * of the reference in the reference's order, so the counts are those of the table widened to f32, bit for bit.  `source`
 * stays f32 -- the queries' own vectors, which the caller widens along with gathering them (blp_gather_triple_vectors; they
 * are a few thousand rows) -- as does everything else.  What the 16-bit copy buys: the reference's Wikidata5M batching
 * (eval_batch_size = 2: 4 queries per pass over 4.6 M rows, scripts/blp-*-wikidata5m.sh:18) is one read of the table per
 * batch, HBM-bound; with block_triples <= batch <= 4 and D = 128 or 256 the passes read the 16-bit table AS IT IS (half the
 * bytes per pass; all passes in one launch: blp_rank_all_batches_native16 says whether a call is such a one).  Any other
 * shape of call -- bound by arithmetic, not by the table read -- ranks a widened f32 copy made inside the call (the workspace
 * holds it: N x D x 4 bytes more; a caller that has the f32 table should rank that one instead; blp_amd.ranking does).
 * Workspace: blp_rank_all_batches_workspace_bytes(...), 256-B aligned. */
size_t blp_rank_all_batches_workspace_bytes(int model, int table_dtype, int64_t N, int D, int64_t ld, int64_t n_triples,
                                            int64_t batch, int64_t block_triples);
int blp_rank_all_batches(int model, const void *table, int table_dtype, int64_t N, int D, int64_t ld, const float *source,
                         int64_t S, int64_t ld_src, const int64_t *fixed_row, const float *rel_emb, int64_t R,
                         const int64_t *rel_id, const int64_t *true_row, int64_t n_triples, int64_t batch,
                         int64_t block_triples, const blp_filter *filter, int32_t *counts, void *workspace,
                         size_t workspace_bytes, int device, void *stream);
int blp_rank_all_batches_native16(int model, int table_dtype, int64_t N, int D, int64_t ld, int64_t n_triples, int64_t batch,
                                  int64_t block_triples);

/* Producer of `source` for big tables: out (2n, D) f32 contiguous, out[t] = the vector of triple t's head, out[n + t] = of
 * its tail (train.py:141-142's `ent_emb[heads]` / `ent_emb[tails]` for the whole set of triples), filled only for the
 * entities whose global row ent2idx[id] (NULL: the id itself) lies in this shard's [row_base, row_base + N) and ZERO
 * otherwise -- so that one all-reduce (sum) over the ranks replicates all 2n vectors exactly (x + 0 = x).  With
 * blp_queries.by_position the queries then index this array.  `table` in any storage type (table_dtype: BLP_DTYPE_*; a 16-bit
 * table: ld % 8 == 0); out stays f32: the vectors are widened exactly. */
int blp_gather_triple_vectors(const int64_t *triples, int64_t n, const int64_t *ent2idx, int64_t ent2idx_len,
                              const void *table, int table_dtype, int64_t N, int D, int64_t ld, int64_t row_base, float *out,
                              int device, void *stream);

/* Measurement aid (bench.py): the NEXT blp_rank_all issued by the calling thread records the two
 * hipEvent_t (created by the caller with timing enabled) on its stream immediately before and after
 * the rank pass (rank_tiles, or pre-pass + refinement), so its duration can be read without a profiler.  One-shot;
 * NULL, NULL cancels.  Has no effect on results. */
int blp_profile_next_rank_kernel(void *start_event, void *stop_event);

/* Measurement aid (bench.py): the number of table passes the first ranking launch of a blp_rank_all_batches call with these
 * sizes covers -- every pass of the call when the streaming kernels take the passes of a reference-batched evaluation
 * (train.py:128-171 with eval_batch_size <= 4) in one launch, else 1 (0: nothing to rank).  What the events of
 * blp_profile_next_rank_kernel bracket divides by it. */
int64_t blp_rank_all_batches_passes_per_launch(int model, int table_dtype, int64_t N, int D, int64_t ld, int64_t n_triples,
                                               int64_t batch, int64_t block_triples);

/* Measurement aid (bench.py's `decided_frac`): what the pre-pass of the LAST blp_rank_all / blp_rank_all_shard call that ran on
 * `workspace` (same model, N, D, q_head, q_tail) left to the exact path.  out[0] = (query, candidate) pairs of the block,
 * out[1] = undecided pairs it listed one by one, out[2] = candidates inside the segments it flagged for wholesale exact
 * re-scoring (a workgroup's list was full), out[3] = the path: 0 = no pre-pass (exact kernels took the block), 1 = TransE
 * v_sad_u16, 2 = bf16 x 3 MFMA, 3 = f32-chain MFMA and 4 = any-width TransE (these two are not counted: out[1] = out[2] = -1).
 * Blocks ranked in several candidate slabs report their last slab.  Waits for `stream`, allocates 32 bytes for the call. */
int blp_rank_all_prepass_stats(int model, int64_t N, int D, int64_t q_head, int64_t q_tail, const void *workspace,
                               size_t workspace_bytes, int64_t out[4], int device, void *stream);

/* The same counts from a DENSE score matrix already in HBM: scores (Q, N) f32 with row stride ld, the
 * true entity given per query as a column index (true_idx, the reference's `true_ents`,
 * utils.py:102) or as a score (true_score); exactly one of the two.  Replaces utils.py:103-105 and the
 * filtered overwrite train.py:159-167 for callers that materialise pred_ents, and serves embedding
 * widths the fused kernels are not compiled for (the matrix then comes from blp_score_fwd). */
int blp_rank_from_scores(const float *scores, int64_t Q, int64_t N, int64_t ld, const int64_t *true_idx,
                         const float *true_score, const int64_t *filt_rowptr, const int64_t *filt_col,
                         int32_t *counts, int device, void *stream);

/* utils.py:104-109 on the counts of blp_rank_all: realistic rank = ((gt + 1) + ge) / 2,
 * rr = 1 / rank (f32), hits = rank <= k.   rr (Q, 2) f32 {raw, filtered};
 * hits (Q, 2, 3) uint8 for k = k_values[0..2] (train.py:72: 1, 3, 10). */
int blp_rank_metrics(const int32_t *counts, int64_t Q, const int32_t k_values[3], float *rr,
                     uint8_t *hits, int device, void *stream);

/* The accumulation of train.py:152-157 on the device: sums[0..1] = sum over queries of the reciprocal
 * rank (raw, filtered), sums[2 + 3 v + j] = number of queries with avg rank <= k_values[j] (v = 0 raw,
 * 1 filtered), all f64, summed in a fixed order (reproducible).  Divide by Q for MRR / Hits@k.
 * `sums` must have room for BLP_METRIC_SUMS_DOUBLES doubles: the 8 results come first, the rest is
 * scratch for the per-block partial sums of large Q. */
#define BLP_METRIC_SUMS_DOUBLES 520
int blp_rank_metric_sums(const int32_t *counts, int64_t Q, const int32_t k_values[3], double *sums,
                         int device, void *stream);

/* --------------------------------------------------------------------------------------------
 * score_fn(heads, tails, rels)   models.py:222-248, any broadcast the reference uses.
 *
 * Output is an (M0, M1) f32 matrix; each operand row is addressed as
 *     base + i0 * stride0 + i1 * stride1      (strides in floats; 0 broadcasts that axis)
 * which covers the eval shapes (1,N,D)x(B,1,D) -> (B,N) of train.py:146-147, the training shapes
 * (B,K,D)x(B,1,D) -> (B,K) of models.py:67 and plain aligned rows (M0 = 1 or M1 = 1).
 * Scores are bit-identical to the reference's torch-CPU result.
 * -------------------------------------------------------------------------------------------- */
int blp_score_fwd(int model, int D, int64_t M0, int64_t M1,
                  const float *heads, int64_t h_s0, int64_t h_s1,
                  const float *tails, int64_t t_s0, int64_t t_s1,
                  const float *rels, int64_t r_s0, int64_t r_s1,
                  float *out, int device, void *stream);
/* d score / d operand for every (i0, i1): grad_* are (M0, M1, D) f32 dense (NULL = skip); the
 * caller reduces over broadcast axes.  grad_out is (M0, M1). */
int blp_score_bwd(int model, int D, int64_t M0, int64_t M1,
                  const float *heads, int64_t h_s0, int64_t h_s1,
                  const float *tails, int64_t t_s0, int64_t t_s1,
                  const float *rels, int64_t r_s0, int64_t r_s1,
                  const float *grad_out, float *grad_heads, float *grad_tails, float *grad_rels,
                  int device, void *stream);

/* --------------------------------------------------------------------------------------------
 * In-batch negatives loss   LinkPrediction.compute_loss, models.py:51-70 (+ :251-266)
 *
 *   ent_embs  (B, 2, D) contiguous == (2B, D): row 2b = head of b, 2b+1 = tail of b
 *   rel_vecs  (B, D)    rel_emb(rels) already gathered (its backward stays in torch)
 *   neg_idx   (B, K, 2) int64, values in [0, 2B): rows of ent_embs.view(2B, D)  (data.py:35-81)
 *   regularizer: models.py:59-60, applied iff > 0
 * Storage types: ent_embs and grad_ent are ent_dtype, rel_vecs and grad_rel are rel_dtype (= ent_dtype, or f32: nn.Embedding
 *      rows stay f32 under autocast -- BASELINE config 5: half-precision embeddings).  Half operands are widened exactly and
 *      every operation is the f32 one of the reference (f32 accumulate); the loss and the saved scores stay f32; gradients
 *      are rounded once on store.  The reference has no half path: parity there is a tolerance against the f32 oracle on the
 *      widened inputs (tests/test_gpu_parity.py::test_inbatch_loss_half_*).
 * fwd: loss (1) f32; save_pos (blp_inbatch_loss_save_floats(model, B, K, D) floats) -- the B positive scores, the
 *      workgroups' partial loss sums, and an INDEX of neg_idx (per chunk of consecutive entries: the entries grouped by
 *      the row they name, in entry order) -- and save_neg (B, K) f32 scores are kept for bwd.  Scores: the bilinear models
 *      32 lanes per pair = torch.sum's 32 running sums, folded by wavefront shuffles; TransE four lanes per pair, the
 *      running L1 sum walking through them -- bit-identical to the reference at the scripts' widths.  Loss: every
 *      workgroup adds its slots' terms in f64 (slot order); the partial sums are then added in workgroup order -- by the
 *      workgroup that takes the last ticket when there are at most 96 of them (ONE launch: the TransE step at the scripts'
 *      batch sizes), else by a second, one-workgroup launch (cheaper than that many tickets).  A fixed order either way.
 *      ticket: BLP_INBATCH_TICKET_INTS int32 the caller keeps PER STREAM, ZERO on entry; the kernel leaves them zero when it
 *      completes (calls on one stream are ordered and may share them; concurrent streams need their own).
 * bwd: ONE launch.  grad_ent (2B, D), grad_rel (B, D), both overwritten, scaled by *grad_loss (device scalar).  O(B K) work:
 *      a row's negatives come from the forward's index, in entry order.  Deterministic (no float atomics).  margin_loss
 *      passes gradient where the hinge is exactly 0 (models.py:252-253 masks in place).
 * Floating point: the scores are the reference's elementwise terms summed in tree order, so the loss agrees with the
 * reference to ~1e-6 relative and the gradients to ~1e-5 (the bit-exact score_fn is blp_score_fwd).
 * -------------------------------------------------------------------------------------------- */
#define BLP_INBATCH_TICKET_INTS 4
size_t blp_inbatch_loss_save_floats(int model, int B, int K, int D);
/* kernel launches of blp_inbatch_loss_fwd at these sizes: 1 (at most 96 scoring workgroups: the last one finishes the loss) or 2 */
int blp_inbatch_loss_fwd_launches(int model, int B, int K, int D, float regularizer);
int blp_inbatch_loss_fwd(int model, int loss, int ent_dtype, int rel_dtype, const void *ent_embs,
                           const void *rel_vecs, const int64_t *neg_idx, int B, int K, int D,
                           float regularizer, float *out_loss, float *save_pos, float *save_neg,
                           int32_t *ticket, int device, void *stream);
int blp_inbatch_loss_bwd(int model, int loss, int ent_dtype, int rel_dtype, const void *ent_embs,
                           const void *rel_vecs, const int64_t *neg_idx, int B, int K, int D,
                           float regularizer, const float *grad_loss, const float *save_pos,
                           const float *save_neg, void *grad_ent, void *grad_rel, int device,
                           void *stream);

/* --------------------------------------------------------------------------------------------
 * The evaluation loop's per-batch prelude for a whole set of n triples, in one kernel (the producer of blp_rank_all's
 * query arguments and of blp_rank_all_ex's filter segments):
 *
 *   heads = [ent2idx[ent] for ent in triples[:, 0]], tails likewise, assert min >= 0        train.py:134-138
 *   head_embs = ent_emb[heads], tail_embs = ent_emb[tails], rel_embs = rel_emb(rels)       train.py:141-145
 *   utils.get_triple_filters(triples, graph, ...)                                           utils.py:46-83
 *
 * Query order: blocks of `block` triples, each block as [its head-replacing queries | its tail-replacing queries]
 * (train.py:149), so that block b is rows [2 b block, ...) of every output and can be handed to blp_rank_all(_ex) as is.
 *   triples      (n, 3) int64 rows (head id, tail id, relation id) -- data.py:128 column order
 *   ent2idx      id -> row of `source`, -1 = not a candidate (utils.make_ent2idx); NULL: ids are rows
 *   source       (src_rows, D) f32, row stride ld: the entity table;  rel_emb (R, D) f32 contiguous
 *   heads_key / tails_key  sorted keys entity * index_R + relation of the filtering graph's (tail, rel) -> heads and
 *                (head, rel) -> tails indices (blp_amd.utils.FilterIndex); both NULL: no filter outputs
 * Outputs (2n rows each): q_fixed, q_rel (2n, D) f32 (both NULL: no vectors are gathered -- blp_rank_all_idx takes
 * fixed_row / rel_ids instead); fixed_row, true_row, rel_ids int64 (fixed_row may be NULL); seg_lo / seg_hi: the
 * query's slice of the caller's value array [heads' values | tails' values] (tail-side slices are offset by n_heads);
 * exclude: the triple's own entity id; *ids_min: 0, or -1 if any id has no row / any relation is outside [0, R) (such
 * queries get zero vectors and rows / relation 0: the caller must check before trusting the counts, as
 * train.py:137-138 asserts).
 * D % 4 == 0; 16-byte aligned source / rel_emb / q_fixed / q_rel; ld % 4 == 0.
 * -------------------------------------------------------------------------------------------- */
typedef struct blp_queries {
    const int64_t *triples; int64_t n, block;
    const int64_t *ent2idx; int64_t ent2idx_len;
    const float *source; int64_t src_rows, ld; int D;
    const float *rel_emb; int64_t R;
    const int64_t *heads_key; int64_t n_heads;
    const int64_t *tails_key; int64_t n_tails;
    int64_t index_R;
    float *q_fixed; float *q_rel; int64_t *true_row; int64_t *rel_ids; int32_t *ids_min;
    int64_t *seg_lo; int64_t *seg_hi; int64_t *exclude;
    int64_t *fixed_row;
    int64_t by_position;     /* 0: fixed_row / true_row are rows of `source` (= ent2idx[id]).  1: `source` is the (2n, D) array
                              * of blp_gather_triple_vectors and fixed_row / true_row are POSITIONS in it (head of triple t = t,
                              * tail = n + t); ent2idx and src_rows (= rows of the whole table) then only serve the id check. */
} blp_queries;
int blp_build_queries(const blp_queries *q, int device, void *stream);

/* --------------------------------------------------------------------------------------------
 * Entity-table build, last step (the producer of blp_rank_all's `table`), for the BERT encoders:
 *
 *   embs = self.enc_linear(embs)              models.py:110-111 (nn.Linear(hidden, dim, bias=False), :104)
 *   ent_emb = F.normalize(ent_emb, dim=-1)    models.py:40-41   (iff normalize != 0: TransE)
 *   ent_emb[idx:idx + batch] = batch_emb      train.py:109-113
 *
 *   out[i, :] = x[i, :] . w^T  [ / max(||.||_2, 1e-12) ]      i < n
 *
 *   x    (n, E) f32, row stride ldx floats (the [CLS] rows of the encoder output are strided: pass the view);
 *   w    (D, E) f32 row-major (enc_linear.weight);  out  (n, D) f32, row stride ldo: rows of the table shard.
 * f32 operands and accumulation on the matrix cores (v_mfma_f32_32x32x2_f32); a floating-point GEMM: results agree
 * with the torch expression to ~1e-6 relative, not bit for bit (the K-reduction order is the kernel's own).
 * blp_project_rows_supported: E % 4 == 0 and D in {64, 128, 256}; x, w, out 16-byte aligned, ldx % 4 == 0.
 * -------------------------------------------------------------------------------------------- */
int blp_project_rows_supported(int E, int D);
int blp_project_rows(const float *x, int64_t n, int64_t ldx, const float *w, int E, int D, int normalize,
                     float *out, int64_t ldo, int device, void *stream);

/* --------------------------------------------------------------------------------------------
 * Entity-table build for the bag-of-words encoder (models.py:143-155; the glove-bow / bert-bow scripts), with the steps
 * that follow it:
 *
 *   embs = self.embeddings(text_tok); lengths = torch.sum(text_mask, dim=-1, keepdim=True)      models.py:150-151
 *   embs = torch.sum(text_mask.unsqueeze(dim=-1) * embs, dim=1) / lengths                       models.py:152-153
 *   ent_emb = F.normalize(ent_emb, dim=-1)    models.py:40-41 (iff normalize != 0)
 *   ent_emb[idx:idx + batch] = batch_emb      train.py:109-113
 *
 *   out[i, :] = (sum_l mask[i, l] * emb[tok[i, l], :]) / (sum_l mask[i, l])  [ / max(||.||_2, 1e-12) ]      i < n
 *
 *   tok   (n, L) int64 token ids, row-major;  mask (n, L) f32 or NULL (all ones: models.py:147-148);
 *   emb   (V, E) f32 row-major (embeddings.weight);  out (n, E) f32, row stride ldo: rows of the table shard;
 *   bad_tok  int32 on the device, set to -1 if a token id was outside [0, V) (row 0 is read for it; nn.Embedding would
 *            raise) and left alone otherwise: the caller initialises it to 0 and reads it when convenient.
 * No (n, L, E) temporary: every gathered row is read once.  Floating point (tokens summed in order, product and sum rounded
 * separately): agrees with the torch expression to ~1e-6 relative, not bit for bit.
 * blp_bow_rows_supported: E % 4 == 0, E <= 1024; emb, out 16-byte aligned, ldo % 4 == 0, ldo >= E.
 * -------------------------------------------------------------------------------------------- */
int blp_bow_rows_supported(int E);
int blp_bow_rows(const int64_t *tok, const float *mask, int64_t n, int L, const float *emb, int64_t V, int E,
                 int normalize, float *out, int64_t ldo, int32_t *bad_tok, int device, void *stream);

/* --------------------------------------------------------------------------------------------
 * Entity-table build for the DKRL encoder (models.py:158-204; the bert-dkrl / glove-dkrl scripts), with the steps that
 * follow it:
 *
 *   embs = self.embeddings(text_tok) * text_mask.unsqueeze(-1)                                    models.py:177
 *   embs = self.conv1(F.pad(embs.transpose(1, 2), [0, 1])) * text_mask          Conv1d(E, dim, 2)  models.py:180-187
 *   embs = F.max_pool1d(embs, 4); text_mask = F.max_pool1d(text_mask, 4)        (L >= 4)           models.py:188-195
 *   embs = self.conv2(F.pad(torch.tanh(embs), [0, 1]))                          Conv1d(dim, dim, 2) models.py:196-198
 *   embs = torch.tanh(torch.sum(embs * text_mask, dim=-1) / lengths)                               models.py:199-202
 *   ent_emb = F.normalize(ent_emb, dim=-1)    models.py:40-41 (iff normalize != 0)
 *   ent_emb[idx:idx + batch] = batch_emb      train.py:109-113
 *
 *   tok (n, L) int64, mask (n, L) f32 or NULL (all ones), emb (V, E) f32 as for blp_bow_rows;
 *   w1 (dim, E, 2), b1 (dim): conv1.weight / .bias;  w2 (dim, dim, 2), b2 (dim): conv2.weight / .bias, contiguous f32;
 *   out (n, dim) f32, row stride ldo: rows of the table shard;  bad_tok as for blp_bow_rows.
 * One kernel, nothing materialised: conv1 on the matrix cores with f32 operands (v_mfma_f32_32x32x2_f32), bias / mask /
 * max-pool / tanh in the accumulator registers, conv2 + masked mean folded into two 128-vectors per entity (they are linear
 * in the pooled activations).  A masked position contributes exact zeros (the stock expression multiplies the gathered
 * row by 0: the same unless the embedding row holds Inf / NaN).  Floating point: agrees with the stock modules to ~1e-6,
 * not bit for bit (the reduction orders differ from the convolution library's).
 * blp_dkrl_rows_supported: E % 4 == 0, dim == 128 (every DKRL script's), 4 <= L <= 64 (max_len 32 / 64; shorter chunks
 * change the pooling window, models.py:188-193, and stay with the stock modules); emb, w1, w2 16-byte aligned, ldo >= dim.
 * -------------------------------------------------------------------------------------------- */
int blp_dkrl_rows_supported(int E, int D, int L);
int blp_dkrl_rows(const int64_t *tok, const float *mask, int64_t n, int L, const float *emb, int64_t V, int E,
                  const float *w1, const float *b1, const float *w2, const float *b2, int D, int normalize, float *out,
                  int64_t ldo, int32_t *bad_tok, int device, void *stream);

/* --------------------------------------------------------------------------------------------
 * Test / A-B hooks -- NOT part of the production library.  They are compiled only with -DBLP_TEST_HOOKS, into a second
 * library (blp_amd/libblp_hip.hooks.so) that tests/ and tools/ load; libblp_hip.so exports neither symbol and has no
 * mutable process-wide state.  The library never reads the environment; the kernel-selection and slab-size overrides the
 * parity tests need (force the exact f32 kernels, the f32-chain GEMM, tiny candidate slabs, ...) are process-wide integer
 * knobs of the hooks build, 0 = automatic (names: blp_amd/csrc/knobs.h).  Results never depend on a knob.
 * -------------------------------------------------------------------------------------------- */
#ifdef BLP_TEST_HOOKS
int blp_debug_set_knob(const char *name, long long value);

/* Test hook for the error band of the bilinear pre-pass: the NEXT blp_rank_all of the calling thread on a DistMult /
 * ComplEx / SimplE block (D = 128, >= 64 queries, one candidate slab) runs the SAME bf16 x 3 MFMA sequence and band
 * arithmetic as always but, instead of deciding, stores the approximate score S~ and the band half-width eps of every
 * (query, candidate) pair into two dense (Q, N) f32 matrices; `counts` of that call are meaningless.  One-shot (the
 * pointers are dropped when that call returns); NULL, NULL cancels. */
int blp_debug_gemm_dump(float *scores, float *eps);

/* Forget `device`'s self-test verdict: the next blp_selftest / bilinear block tests again (with knob "mfma_selftest" = 1 it
 * then reports a violation whatever it measures -- how tests reach the f32-chain route). */
int blp_debug_reset_selftest(int device);
#endif

#ifdef __cplusplus
}
#endif
#endif /* BLP_HIP_H */
