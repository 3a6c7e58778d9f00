/*
 * blp_oracle.c -- CPU ORACLE for the BLP link-prediction scoring / ranking hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product path (blp_amd/) never does.
 *
 * It is a plain-C, scalar, order-exact restatement of the arithmetic the reference
 * (dfdazac/blp, Python/PyTorch) performs on the torch CPU backend:
 *
 *   transe_score     /root/reference/models.py:222-223   -||h + r - t||_1
 *   distmult_score   /root/reference/models.py:226-227   sum (h*r)*t
 *   complex_score    /root/reference/models.py:230-239   4-term complex product, sum
 *   simple_score     /root/reference/models.py:242-248   2-term SimplE, sum, /2
 *   get_metrics      /root/reference/utils.py:86-111     rank counts (pred > true, pred >= true)
 *   filtered re-rank /root/reference/train.py:159-171    pred[mask] = pred.min() - 1
 *
 * The reduction ORDER is part of the contract (ranks must be bit-identical):
 *   - torch.norm(x, p=1, dim=-1) on CPU is a strict left-to-right f32 sum
 *     (ATen norm kernel -> binary_kernel_reduce with AbsSumOps, no vectorisation for p=1).
 *   - torch.sum(x, dim=-1) on CPU over a contiguous inner dim of n floats is
 *     ATen/native/cpu/SumKernel.cpp vectorized_inner_sum -> row_sum -> multi_row_sum:
 *     8-lane vectors (the AVX2 kernel is the one dispatched, also on AVX-512 hosts),
 *     4 ILP accumulators, cascade levels of 16 chunks, scalar tail added first,
 *     then the 8 lanes added left to right.  torch_inner_sum() below restates it for any n.
 *   Both were checked bit-for-bit against the imported reference functions in the build
 *   container (tests/golden/make_golden.py, tests/test_oracle_golden.py).
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math -shared -fPIC (see oracle/Makefile).
 * -ffp-contract=off matters: an FMA would fuse (h*r)*t + acc and change the last bit.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

enum { BLP_TRANSE = 0, BLP_DISTMULT = 1, BLP_COMPLEX = 2, BLP_SIMPLE = 3 };
enum { SIDE_HEAD = 0, SIDE_TAIL = 1 }; /* which position the candidates replace */

/* ---- torch.sum(dim=-1) order (SumKernel.cpp: vectorized_inner_sum / row_sum / multi_row_sum) ---- */
static int ceil_log2_i64(int64_t x) {
    if (x <= 2) return 1;
    int l = 0;
    int64_t v = x - 1;
    while (v > 0) { v >>= 1; ++l; }
    return l;
}

float blp_oracle_torch_inner_sum(const float *p, int64_t n) {
    enum { VEC = 8, ILP = 4, LEVELS = 4 };
    const int64_t vec_size = n / VEC;        /* number of 8-lane vectors              */
    const int64_t size_ilp = vec_size / ILP; /* number of 4-vector "rows" of the ILP  */
    float acc[LEVELS][ILP][VEC];
    memset(acc, 0, sizeof(acc));

    /* multi_row_sum over size_ilp rows of ILP vectors, cascade every level_step rows */
    int level_power = ceil_log2_i64(size_ilp) / LEVELS;
    if (level_power < 4) level_power = 4;
    const int64_t level_step = (int64_t)1 << level_power;
    const int64_t level_mask = level_step - 1;
    int64_t i = 0;
    for (; i + level_step <= size_ilp;) {
        for (int64_t j = 0; j < level_step; ++j, ++i)
            for (int k = 0; k < ILP; ++k)
                for (int l = 0; l < VEC; ++l)
                    acc[0][k][l] += p[(i * ILP + k) * VEC + l];
        for (int j = 1; j < LEVELS; ++j) {
            for (int k = 0; k < ILP; ++k)
                for (int l = 0; l < VEC; ++l) {
                    acc[j][k][l] += acc[j - 1][k][l];
                    acc[j - 1][k][l] = 0.0f;
                }
            const int64_t mask = level_mask << (j * level_power);
            if ((i & mask) != 0) break;
        }
    }
    for (; i < size_ilp; ++i)
        for (int k = 0; k < ILP; ++k)
            for (int l = 0; l < VEC; ++l)
                acc[0][k][l] += p[(i * ILP + k) * VEC + l];
    for (int j = 1; j < LEVELS; ++j)
        for (int k = 0; k < ILP; ++k)
            for (int l = 0; l < VEC; ++l)
                acc[0][k][l] += acc[j][k][l];

    /* row_sum: leftover whole vectors go to partial 0, then partials 1..3 fold into 0 */
    for (int64_t v = size_ilp * ILP; v < vec_size; ++v)
        for (int l = 0; l < VEC; ++l)
            acc[0][0][l] += p[v * VEC + l];
    for (int k = 1; k < ILP; ++k)
        for (int l = 0; l < VEC; ++l)
            acc[0][0][l] += acc[0][k][l];

    /* vectorized_inner_sum: scalar tail first, then the 8 lanes left to right */
    float final_acc = 0.0f;
    for (int64_t k = vec_size * VEC; k < n; ++k) final_acc += p[k];
    for (int l = 0; l < VEC; ++l) final_acc += acc[0][0][l];
    return final_acc;
}

/* ---- the four score functions, one (h, t, r) triple of D floats each ---- */
static float transe_one(const float *h, const float *t, const float *r, int D) {
    float acc = 0.0f; /* models.py:223: (h + r) - t, abs, strict sequential sum, negate */
    for (int d = 0; d < D; ++d) {
        float x = h[d] + r[d];
        x = x - t[d];
        acc = acc + fabsf(x);
    }
    return -acc;
}

static float distmult_one(const float *h, const float *t, const float *r, int D, float *tmp) {
    for (int d = 0; d < D; ++d) { /* models.py:227: (h * r) * t, each product rounded */
        float x = h[d] * r[d];
        tmp[d] = x * t[d];
    }
    return blp_oracle_torch_inner_sum(tmp, D);
}

static float complex_one(const float *h, const float *t, const float *r, int D, float *tmp) {
    const int H = D / 2; /* models.py:231-239 */
    const float *hr = h, *hi = h + H, *tr = t, *ti = t + H, *rr = r, *ri = r + H;
    for (int j = 0; j < H; ++j) {
        float a = rr[j] * hr[j]; a = a * tr[j];
        float b = rr[j] * hi[j]; b = b * ti[j];
        float c = ri[j] * hr[j]; c = c * ti[j];
        float d = ri[j] * hi[j]; d = d * tr[j];
        float s = a + b;
        s = s + c;
        tmp[j] = s - d;
    }
    return blp_oracle_torch_inner_sum(tmp, H);
}

static float simple_one(const float *h, const float *t, const float *r, int D, float *tmp) {
    const int H = D / 2; /* models.py:243-248 */
    const float *hh = h, *ht = h + H, *th = t, *tt = t + H, *ra = r, *rb = r + H;
    for (int j = 0; j < H; ++j) {
        float a = hh[j] * ra[j]; a = a * tt[j];
        float b = th[j] * rb[j]; b = b * ht[j];
        tmp[j] = a + b;
    }
    return blp_oracle_torch_inner_sum(tmp, H) / 2.0f;
}

float blp_oracle_score_one(int model, const float *h, const float *t, const float *r, int D) {
    float stack_tmp[1024];
    float *tmp = D <= 1024 ? stack_tmp : (float *)malloc(sizeof(float) * (size_t)D);
    float s;
    switch (model) {
    case BLP_TRANSE:   s = transe_one(h, t, r, D); break;
    case BLP_DISTMULT: s = distmult_one(h, t, r, D, tmp); break;
    case BLP_COMPLEX:  s = complex_one(h, t, r, D, tmp); break;
    case BLP_SIMPLE:   s = simple_one(h, t, r, D, tmp); break;
    default:           s = NAN; break;
    }
    if (tmp != stack_tmp) free(tmp);
    return s;
}

/* score_fn over M aligned (h, t, r) rows: out[m] = score(h[m], t[m], r[m]) */
void blp_oracle_score_pairs(int model, const float *h, const float *t, const float *r,
                            int64_t M, int D, float *out) {
    for (int64_t m = 0; m < M; ++m)
        out[m] = blp_oracle_score_one(model, h + m * D, t + m * D, r + m * D, D);
}

/* train.py:146-147: score every table row as replacement head (side 0) or tail (side 1).
 * out is (Q, N) row-major. */
void blp_oracle_score_all(int model, int side, const float *table, int64_t N, int D, int64_t ld,
                          const float *q_fixed, const float *q_rel, int64_t Q, float *out) {
#pragma omp parallel for schedule(dynamic, 4)
    for (int64_t q = 0; q < Q; ++q) {
        const float *f = q_fixed + q * D, *r = q_rel + q * D;
        for (int64_t n = 0; n < N; ++n) {
            const float *e = table + n * ld;
            out[q * N + n] = side == SIDE_HEAD ? blp_oracle_score_one(model, e, f, r, D)
                                               : blp_oracle_score_one(model, f, e, r, D);
        }
    }
}

/* utils.py:103-105 on one row of scores: counts[0] = #(pred > true), counts[1] = #(pred >= true) */
void blp_oracle_count_row(const float *scores, int64_t N, float true_score, int32_t *gt, int32_t *ge) {
    int32_t a = 0, b = 0;
    for (int64_t n = 0; n < N; ++n) {
        a += scores[n] > true_score;
        b += scores[n] >= true_score;
    }
    *gt = a;
    *ge = b;
}

/*
 * Fused restatement of train.py:146-171 + utils.py:103-105 for Q queries of one side.
 * counts[q] = {gt, ge, gt_filt, ge_filt}.
 *   true score: either table[true_row[q]] (true_row != NULL) or q_true + q*D (sharded ranking:
 *   the true entity's vector is replicated, the row may live on another shard).
 *   filter: CSR (filt_rowptr[Q+1], filt_col[nnz]) of table rows that train.py:165 overwrites with
 *   pred.min() - 1.0.  Those rows can never be > or >= the true score, so the filtered counts are
 *   the counts over the rows NOT listed (the true entity is never listed: utils.py:71,78).
 *   filt_rowptr == NULL -> filtered counts equal the raw counts.
 */
void blp_oracle_rank_counts(int model, int side, const float *table, int64_t N, int D, int64_t ld,
                            const float *q_fixed, const float *q_rel, const int64_t *true_row,
                            const float *q_true, int64_t Q, const int64_t *filt_rowptr,
                            const int64_t *filt_col, int32_t *counts) {
    /* queries are independent: with -fopenmp (oracle/Makefile) they are spread over the host cores, which
     * changes nothing in any query's arithmetic (full-size parity tests check thousands of queries) */
#pragma omp parallel for schedule(dynamic, 4)
    for (int64_t q = 0; q < Q; ++q) {
        const float *f = q_fixed + q * D, *r = q_rel + q * D;
        const float *te = true_row ? table + true_row[q] * ld : q_true + q * D;
        const float ts = side == SIDE_HEAD ? blp_oracle_score_one(model, te, f, r, D)
                                           : blp_oracle_score_one(model, f, te, r, D);
        int32_t gt = 0, ge = 0;
        for (int64_t n = 0; n < N; ++n) {
            const float *e = table + n * ld;
            const float s = side == SIDE_HEAD ? blp_oracle_score_one(model, e, f, r, D)
                                              : blp_oracle_score_one(model, f, e, r, D);
            gt += s > ts;
            ge += s >= ts;
        }
        int32_t fgt = 0, fge = 0;
        if (filt_rowptr) {
            for (int64_t k = filt_rowptr[q]; k < filt_rowptr[q + 1]; ++k) {
                const float *e = table + filt_col[k] * ld;
                const float s = side == SIDE_HEAD ? blp_oracle_score_one(model, e, f, r, D)
                                                  : blp_oracle_score_one(model, f, e, r, D);
                fgt += s > ts;
                fge += s >= ts;
            }
        }
        counts[4 * q + 0] = gt;
        counts[4 * q + 1] = ge;
        counts[4 * q + 2] = gt - fgt;
        counts[4 * q + 3] = ge - fge;
    }
}

/* utils.py:104-109: realistic rank = (best + worst) / 2 with best = gt + 1, worst = ge;
 * reciprocal in f32; hits = avg <= k.  hits is (Q, nk) bytes. */
void blp_oracle_metrics_from_counts(const int32_t *gt, const int32_t *ge, int64_t Q, int64_t stride,
                                    const int32_t *k_values, int nk, float *rr, uint8_t *hits) {
    for (int64_t q = 0; q < Q; ++q) {
        const int64_t best = (int64_t)gt[q * stride] + 1, worst = ge[q * stride];
        const float avg = (float)(best + worst) * 0.5f;
        rr[q] = 1.0f / avg;
        for (int j = 0; j < nk; ++j) hits[q * nk + j] = avg <= (float)k_values[j];
    }
}
