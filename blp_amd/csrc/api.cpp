// api.cpp -- the C-ABI of libblp_hip.so (include/blp_hip.h): argument checking, device selection,
// error reporting and dispatch to the launchers in the .hip files.  No kernel code here.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstring>

#include "../../include/blp_hip.h"
#include "knobs.h"
#include "launch.h"
#include "rank_common.h"

namespace {

thread_local char g_error[512] = "";
thread_local hipEvent_t g_prof_start = nullptr, g_prof_stop = nullptr;  // blp_profile_next_rank_kernel

int fail(int status, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof(g_error), fmt, ap);
    va_end(ap);
    return status;
}

int hip_fail(hipError_t err, const char* what) {
    return fail(BLP_ERR_HIP, "%s: %s", what, hipGetErrorString(err));
}

// Make `device` current for the duration of a call, restore the caller's device afterwards
// (torch keeps its own notion of the current device per thread).
class DeviceGuard {
  public:
    explicit DeviceGuard(int device) {
        err_ = hipGetDevice(&prev_);
        if (err_ == hipSuccess && prev_ != device) {
            err_ = hipSetDevice(device);
            switched_ = err_ == hipSuccess;
        }
    }
    ~DeviceGuard() {
        if (switched_) (void)hipSetDevice(prev_);
    }
    hipError_t error() const { return err_; }

  private:
    int prev_ = 0;
    bool switched_ = false;
    hipError_t err_ = hipSuccess;
};

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
bool valid_model(int m) { return m >= BLP_TRANSE && m <= BLP_SIMPLE; }

int compute_units(int device, int* out) {
    // Cached per device: hipDeviceGetAttribute is cheap but not free, and blp_rank_all is called
    // per query block.  Plain ints written once; a race only repeats the query.
    static int cached[64] = {0};
    if (device >= 0 && device < 64 && cached[device] > 0) {
        *out = cached[device];
        return BLP_OK;
    }
    int cu = 0;
    hipError_t err = hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, device);
    if (err != hipSuccess) return hip_fail(err, "hipDeviceGetAttribute(MultiprocessorCount)");
    if (device >= 0 && device < 64) cached[device] = cu;
    *out = cu;
    return BLP_OK;
}

blp::StridedRows rows(const float* p, int64_t s0, int64_t s1) { return blp::StridedRows{p, s0, s1}; }

// score_fn operands: D floats readable at every addressed row.  Any width (the reference takes any `dim`): the
// torch.sum order of the bilinear models runs from registers at the scripts' widths (reduction width a multiple
// of 32 below 512) and through the general, slower routine otherwise (score_direct.h).  ComplEx / SimplE split
// the vector into halves.
int check_score_dim(int model, int D) {
    if (D <= 0) return fail(BLP_ERR_BAD_ARG, "D must be positive (got %d)", D);
    if ((model == BLP_COMPLEX || model == BLP_SIMPLE) && D % 2)
        return fail(BLP_ERR_UNSUPPORTED_DIM, "model %d splits the embedding into halves: D = %d must be even", model, D);
    return BLP_OK;
}

#ifdef BLP_TEST_HOOKS
std::atomic<long long> g_knobs[blp::KNOB_COUNT];  // zero-initialised: every knob automatic
const char* const kKnobNames[blp::KNOB_COUNT] = {"rank_kernel", "gemm_kernel", "sad_queries_per_group", "sad_pass_groups",
                                                 "sad_min_queries", "gemm_pass_words", "gemm_tiles_per_chunk",
                                                 "exact_query_chunk", "small_kernel",
                                                 "stream_kernel", "dkrl_split", "mfma_selftest", "inbatch_probe", "inbatch_shares"};
#endif

}  // namespace

#ifdef BLP_TEST_HOOKS
namespace blp {
long long knob(int which) { return g_knobs[which].load(std::memory_order_relaxed); }
}  // namespace blp
#endif

extern "C" {

#ifdef BLP_TEST_HOOKS
int blp_debug_set_knob(const char* name, long long value) {
    if (!name) return fail(BLP_ERR_BAD_ARG, "blp_debug_set_knob: NULL name");
    for (int i = 0; i < blp::KNOB_COUNT; ++i)
        if (std::strcmp(name, kKnobNames[i]) == 0) {
            g_knobs[i].store(value, std::memory_order_relaxed);
            return BLP_OK;
        }
    return fail(BLP_ERR_BAD_ARG, "blp_debug_set_knob: unknown knob '%s'", name);
}

int blp_debug_reset_selftest(int device) {
    blp::mfma_accum_reset(device);
    return BLP_OK;
}

int blp_debug_gemm_dump(float* scores, float* eps) {
    if ((scores == nullptr) != (eps == nullptr)) return fail(BLP_ERR_BAD_ARG, "blp_debug_gemm_dump: give both matrices or neither");
    blp::gemm_set_dump(scores, eps);
    return BLP_OK;
}
#endif

int blp_build_queries(const blp_queries* q, int device, void* stream) {
    if (!q) return fail(BLP_ERR_BAD_ARG, "blp_build_queries: NULL argument block");
    if (q->n < 0 || q->block <= 0 || q->D <= 0 || (q->D & 3) || q->src_rows < 0 || q->R < 0 || q->n > (1ll << 40))
        return fail(BLP_ERR_BAD_ARG, "blp_build_queries: bad sizes (n=%lld block=%lld D=%d)", (long long)q->n, (long long)q->block, q->D);
    if (!q->ids_min) return fail(BLP_ERR_BAD_ARG, "blp_build_queries: ids_min is NULL");
    if (q->n > 0 && (!q->triples || !q->source || !q->rel_emb || !q->true_row || !q->rel_ids))
        return fail(BLP_ERR_BAD_ARG, "blp_build_queries: NULL pointer");
    if ((q->q_fixed == nullptr) != (q->q_rel == nullptr))
        return fail(BLP_ERR_BAD_ARG, "blp_build_queries: q_fixed and q_rel are given together or not at all");
    if (!aligned16(q->source) || !aligned16(q->rel_emb) || !aligned16(q->q_fixed) || !aligned16(q->q_rel) || (q->ld & 3) || q->ld < q->D)
        return fail(BLP_ERR_BAD_ARG, "blp_build_queries: source / rel_emb / q_fixed / q_rel must be 16-byte aligned, ld %% 4 == 0, ld >= D");
    const bool with_index = q->heads_key || q->tails_key || q->seg_lo || q->seg_hi || q->exclude;
    if (with_index && (!q->seg_lo || !q->seg_hi || !q->exclude || q->index_R <= 0 || q->n_heads < 0 || q->n_tails < 0 ||
                       (q->n_heads > 0 && !q->heads_key) || (q->n_tails > 0 && !q->tails_key)))
        return fail(BLP_ERR_BAD_ARG, "blp_build_queries: a filter index needs both key arrays, index_R > 0 and all three segment outputs");
    DeviceGuard guard(device);
    if (guard.error() != hipSuccess) return hip_fail(guard.error(), "hipSetDevice");
    blp::QueryBuild a{q->triples, q->n, q->block, q->ent2idx, q->ent2idx_len, q->source, q->src_rows, q->ld, q->D,
                      q->rel_emb, q->R, q->heads_key, q->n_heads, q->tails_key, q->n_tails, q->index_R, q->q_fixed, q->q_rel,
                      q->true_row, q->rel_ids, q->ids_min, with_index ? q->seg_lo : nullptr, q->seg_hi, q->exclude, q->fixed_row,
                      q->by_position};
    hipError_t err = blp::launch_build_queries(a, static_cast<hipStream_t>(stream));
    if (err != hipSuccess) return hip_fail(err, "blp_build_queries launch");
    return BLP_OK;
}

static bool valid_table_dtype(int dtype) { return dtype == BLP_DTYPE_F32 || dtype == BLP_DTYPE_F16 || dtype == BLP_DTYPE_BF16; }
// rows of a table of this storage type start on 16-byte boundaries
static bool table_rows_aligned(const void* table, int dtype, int64_t ld) {
    return aligned16(table) && (ld & (dtype == BLP_DTYPE_F32 ? 3 : 7)) == 0;
}

int blp_gather_triple_vectors(const int64_t* triples, int64_t n, const int64_t* ent2idx, int64_t ent2idx_len, const void* table,
                                int table_dtype, int64_t N, int D, int64_t ld, int64_t row_base, float* out, int device, void* stream) {
    if (!valid_table_dtype(table_dtype)) return fail(BLP_ERR_BAD_ARG, "blp_gather_triple_vectors: unknown table dtype %d", table_dtype);
    if (n < 0 || N < 0 || D <= 0 || (D & 3) || ld < D || n > (1ll << 40) || (ent2idx && ent2idx_len < 0))
        return fail(BLP_ERR_BAD_ARG, "blp_gather_triple_vectors: bad sizes (n=%lld N=%lld D=%d ld=%lld)", (long long)n, (long long)N, D, (long long)ld);
    if (n == 0) return BLP_OK;
    if (!triples || !out || (N > 0 && !table)) return fail(BLP_ERR_BAD_ARG, "blp_gather_triple_vectors: NULL pointer");
    if (!table_rows_aligned(table, table_dtype, ld) || !aligned16(out))
        return fail(BLP_ERR_BAD_ARG, "blp_gather_triple_vectors: table rows / out must be 16-byte aligned (ld %% 4 == 0; a 16-bit table: ld %% 8 == 0)");
    DeviceGuard guard(device);
    if (guard.error() != hipSuccess) return hip_fail(guard.error(), "hipSetDevice");
    hipError_t err = blp::launch_gather_triple_vectors(triples, n, ent2idx, ent2idx_len, table, table_dtype, N, D, ld, row_base, out,
                                                       static_cast<hipStream_t>(stream));
    if (err != hipSuccess) return hip_fail(err, "blp_gather_triple_vectors launch");
    return BLP_OK;
}

int blp_project_rows_supported(int E, int D) { return blp::project_rows_supported(E, D) ? 1 : 0; }

int blp_project_rows(const float* x, int64_t n, int64_t ldx, const float* w, int E, int D, int normalize, float* out,
                     int64_t ldo, int device, void* stream) {
    if (n < 0) return fail(BLP_ERR_BAD_ARG, "blp_project_rows: negative row count");
    if (!blp::project_rows_supported(E, D))
        return fail(BLP_ERR_UNSUPPORTED_DIM, "blp_project_rows: E=%d (needs E %% 4 == 0), D=%d (needs 64 / 128 / 256)", E, D);
    if (n == 0) return BLP_OK;
    if (!x || !w || !out) return fail(BLP_ERR_BAD_ARG, "blp_project_rows: NULL pointer");
    if (!aligned16(x) || !aligned16(w) || !aligned16(out) || (ldx & 3) || ldx < E || ldo < D)
        return fail(BLP_ERR_BAD_ARG, "blp_project_rows: x / w / out must be 16-byte aligned, ldx %% 4 == 0, ldx >= E, ldo >= D");
    DeviceGuard guard(device);
    if (guard.error() != hipSuccess) return hip_fail(guard.error(), "hipSetDevice");
    hipError_t err = blp::launch_project_rows(x, n, ldx, w, E, D, normalize, out, ldo, static_cast<hipStream_t>(stream));
    if (err != hipSuccess) return hip_fail(err, "blp_project_rows launch");
    return BLP_OK;
}

int blp_bow_rows_supported(int E) { return blp::bow_rows_supported(E) ? 1 : 0; }

int blp_bow_rows(const int64_t* tok, const float* mask, int64_t n, int L, const float* emb, int64_t V, int E, int normalize,
                 float* out, int64_t ldo, int32_t* bad_tok, int device, void* stream) {
    if (n < 0 || L < 0 || V < 0) return fail(BLP_ERR_BAD_ARG, "blp_bow_rows: negative size");
    if (!blp::bow_rows_supported(E)) return fail(BLP_ERR_UNSUPPORTED_DIM, "blp_bow_rows: E = %d (needs E %% 4 == 0, E <= 1024)", E);
    if (n == 0) return BLP_OK;
    if (!out || !bad_tok || (L > 0 && (!tok || !emb || V == 0))) return fail(BLP_ERR_BAD_ARG, "blp_bow_rows: NULL pointer or empty embedding table");
    if (!aligned16(emb) || !aligned16(out) || (ldo & 3) || ldo < E)
        return fail(BLP_ERR_BAD_ARG, "blp_bow_rows: emb / out must be 16-byte aligned, ldo %% 4 == 0, ldo >= E");
    DeviceGuard guard(device);
    if (guard.error() != hipSuccess) return hip_fail(guard.error(), "hipSetDevice");
    hipError_t err = blp::launch_bow_rows(tok, mask, n, L, emb, V, E, normalize, out, ldo, bad_tok, static_cast<hipStream_t>(stream));
    if (err != hipSuccess) return hip_fail(err, "blp_bow_rows launch");
    return BLP_OK;
}

int blp_dkrl_rows_supported(int E, int D, int L) { return blp::dkrl_rows_supported(E, D, L) ? 1 : 0; }

int blp_dkrl_rows(const int64_t* tok, const float* mask, int64_t n, int L, const float* emb, int64_t V, int E, const float* w1,
                  const float* b1, const float* w2, const float* b2, int D, int normalize, float* out, int64_t ldo, int32_t* bad_tok,
                  int device, void* stream) {
    if (n < 0 || V < 0) return fail(BLP_ERR_BAD_ARG, "blp_dkrl_rows: negative size");
    if (!blp::dkrl_rows_supported(E, D, L))
        return fail(BLP_ERR_UNSUPPORTED_DIM, "blp_dkrl_rows: E = %d (needs E %% 4 == 0), dim = %d (needs 128), L = %d (needs 4 .. 64)", E, D, L);
    if (n == 0) return BLP_OK;
    if (!tok || !emb || !w1 || !b1 || !w2 || !b2 || !out || !bad_tok || V == 0)
        return fail(BLP_ERR_BAD_ARG, "blp_dkrl_rows: NULL pointer or empty embedding table");
    if (!aligned16(emb) || !aligned16(w1) || !aligned16(w2) || ldo < D)
        return fail(BLP_ERR_BAD_ARG, "blp_dkrl_rows: emb / w1 / w2 must be 16-byte aligned, ldo >= dim");
    DeviceGuard guard(device);
    if (guard.error() != hipSuccess) return hip_fail(guard.error(), "hipSetDevice");
    int cu = 0;
    if (int rc = compute_units(device, &cu)) return rc;
    hipError_t err = blp::launch_dkrl_rows(tok, mask, n, L, emb, V, E, w1, b1, w2, b2, normalize, out, ldo, bad_tok, cu,
                                           static_cast<hipStream_t>(stream));
    if (err != hipSuccess) return hip_fail(err, "blp_dkrl_rows launch");
    return BLP_OK;
}

int blp_version(void) { return BLP_HIP_VERSION; }

const char* blp_last_error(void) { return g_error; }

int blp_device_caps(int device, blp_caps* out) {
    if (!out) return fail(BLP_ERR_BAD_ARG, "blp_device_caps: out is NULL");
    hipDeviceProp_t prop;
    hipError_t err = hipGetDeviceProperties(&prop, device);
    if (err != hipSuccess) return hip_fail(err, "hipGetDeviceProperties");
    out->compute_units = prop.multiProcessorCount;
    out->wavefront_size = prop.warpSize;
    out->lds_bytes_per_cu = (int)prop.maxSharedMemoryPerMultiProcessor;
    out->clock_mhz = prop.clockRate / 1000;
    out->hbm_bytes = (int64_t)prop.totalGlobalMem;
    std::memset(out->arch, 0, sizeof(out->arch));
    std::strncpy(out->arch, prop.gcnArchName, sizeof(out->arch) - 1);
    out->mfma_bf16_accum = blp::mfma_accum_state(device);
    out->mfma_bf16_accum_worst = blp::mfma_accum_worst(device);
    return BLP_OK;
}

int blp_rank_all_prepass_stats(int model, int64_t N, int D, int64_t q_head, int64_t q_tail, const void* workspace,
                               size_t workspace_bytes, int64_t out[4], int device, void* stream) {
    if (!valid_model(model) || !out || N < 0 || q_head < 0 || q_tail < 0 || !blp_rank_all_supported(model, D, q_head, q_tail))
        return fail(BLP_ERR_BAD_ARG, "blp_rank_all_prepass_stats: bad arguments");
    if (!workspace || workspace_bytes < blp_rank_all_workspace_bytes(model, N, D, q_head, q_tail))
        return fail(BLP_ERR_WORKSPACE, "blp_rank_all_prepass_stats: not the workspace of a blp_rank_all call of this shape");
    DeviceGuard guard(device);
    if (guard.error() != hipSuccess) return hip_fail(guard.error(), "hipSetDevice");
    blp::PrepassStats st;
    hipError_t err = blp::prepass_stats(model, D, N, q_head, q_tail, workspace, &st, static_cast<hipStream_t>(stream));
    if (err != hipSuccess) return hip_fail(err, "blp_rank_all_prepass_stats");
    out[0] = st.pairs; out[1] = st.listed; out[2] = st.flagged_rows; out[3] = st.path;
    return BLP_OK;
}

int blp_selftest(int device, void* stream) {
    if (int state = blp::mfma_accum_state(device)) return state;
    DeviceGuard guard(device);
    if (guard.error() != hipSuccess) return hip_fail(guard.error(), "hipSetDevice");
    void* scratch = nullptr;
    hipError_t err = hipMalloc(&scratch, 4096);
    if (err != hipSuccess) return hip_fail(err, "blp_selftest: hipMalloc");
    const int state = blp::mfma_accum_selftest(device, scratch, static_cast<hipStream_t>(stream), &err);
    (void)hipFree(scratch);
    if (err != hipSuccess) return hip_fail(err, "blp_selftest");
    if (state == 0) return fail(BLP_ERR_BAD_ARG, "blp_selftest: the stream is being captured into a graph (or device %d is out of range): no verdict", device);
    return state;
}

int blp_dim_supported(int model, int D) {
    if (!valid_model(model)) return 0;
    if (D != 64 && D != 128 && D != 256) return 0;
    return 1;
}

int blp_rank_all_supported(int model, int D, int64_t q_head, int64_t q_tail) {
    if (!valid_model(model) || q_head < 0 || q_tail < 0) return 0;
    return blp_dim_supported(model, D) || blp::rank_sad_wide_applicable(model, D, q_head, q_tail);
}

size_t blp_rank_all_workspace_bytes(int model, int64_t N, int D, int64_t q_head, int64_t q_tail) {
    if (D <= 0 || N < 0 || q_head < 0 || q_tail < 0) return 0;
    return blp::rank_all_workspace_bytes(model, D, N, q_head, q_tail);
}

}  // extern "C"

// Shared by blp_rank_all (dense query vectors) and blp_rank_all_shard (queries as rows of `source` / of rel_emb)
static int rank_all_checked(int model, const float* table, int64_t N, int D, int64_t ld, const blp::QRows& q_fixed,
                            const blp::QRows& q_rel, const blp::QRows& q_true,
                            int64_t q_head, int64_t q_tail, const blp_filter* filter, int32_t* counts, void* workspace,
                            size_t workspace_bytes, int device, void* stream) {
    if (!valid_model(model)) return fail(BLP_ERR_BAD_ARG, "blp_rank_all: unknown model %d", model);
    if (!blp_rank_all_supported(model, D, q_head, q_tail))
        return fail(BLP_ERR_UNSUPPORTED_DIM,
                    "blp_rank_all: D = %d not supported for this block (64 / 128 / 256 always; TransE at any "
                    "D %% 4 == 0 up to 1024): see blp_rank_all_supported", D);
    if (N < 0 || q_head < 0 || q_tail < 0 || ld < D)
        return fail(BLP_ERR_BAD_ARG, "blp_rank_all: negative size or ld < D (N=%lld q_head=%lld q_tail=%lld ld=%lld)",
                    (long long)N, (long long)q_head, (long long)q_tail, (long long)ld);
    const int64_t Q = q_head + q_tail;
    if (Q == 0) return BLP_OK;
    // counts are int32, the two accumulators of a query share one 64-bit word and pair lists hold 32-bit rows
    if (Q > (1ll << 30) || N >= (1ll << 31))
        return fail(BLP_ERR_BAD_ARG, "blp_rank_all: Q > 2^30 or N >= 2^31 (counts are int32): split the query block / "
                                     "shard the candidate axis");
    if (!q_fixed.base || !q_rel.base || !counts) return fail(BLP_ERR_BAD_ARG, "blp_rank_all: NULL q_fixed / q_rel / counts");
    if (N > 0 && !table) return fail(BLP_ERR_BAD_ARG, "blp_rank_all: NULL table");
    if (!q_true.base) return fail(BLP_ERR_BAD_ARG, "blp_rank_all: exactly one of true_row / q_true must be given");
    blp::FilterSpec spec;
    if (filter) {
        if (!filter->seg_lo || !filter->seg_hi || !filter->values)
            return fail(BLP_ERR_BAD_ARG, "blp_rank_all: filter needs seg_lo, seg_hi and values");
        if (filter->ent2idx && filter->ent2idx_len < 0)
            return fail(BLP_ERR_BAD_ARG, "blp_rank_all: filter ent2idx_len is negative");
        spec.lo = filter->seg_lo; spec.hi = filter->seg_hi; spec.val = filter->values; spec.exclude = filter->exclude;
        spec.ent2idx = filter->ent2idx; spec.ent2idx_len = filter->ent2idx ? filter->ent2idx_len : 0;
        spec.row_base = filter->row_base;
    }
    if (!aligned16(table) || (ld & 3) || !aligned16(q_true.base) || (q_true.ld & 3) || !aligned16(counts) ||
        !aligned16(q_fixed.base) || !aligned16(q_rel.base) || (q_fixed.ld & 3) || (q_rel.ld & 3) || (D & 3))
        return fail(BLP_ERR_BAD_ARG, "blp_rank_all: table / q_fixed / q_rel / q_true / counts must be 16-byte aligned, "
                                     "ld %% 4 == 0 and D %% 4 == 0");
    const size_t need = blp::rank_all_workspace_bytes(model, D, N, q_head, q_tail);
    if (!workspace || workspace_bytes < need || (reinterpret_cast<uintptr_t>(workspace) & 255u))
        return fail(BLP_ERR_WORKSPACE, "blp_rank_all: workspace must be 256-byte aligned and >= %zu bytes (got %zu)",
                    need, workspace_bytes);
    DeviceGuard guard(device);
    if (guard.error() != hipSuccess) return hip_fail(guard.error(), "hipSetDevice");
    int cu = 0;
    if (int rc = compute_units(device, &cu)) return rc;
    hipEvent_t ev0 = g_prof_start, ev1 = g_prof_stop;
    g_prof_start = g_prof_stop = nullptr;
    hipError_t err = blp::launch_rank_all(model, D, table, N, ld, q_fixed, q_rel, q_true, q_head, q_tail,
                                          spec, counts, workspace, cu, static_cast<hipStream_t>(stream), ev0, ev1);
    if (err != hipSuccess) return hip_fail(err, "blp_rank_all launch");
    return BLP_OK;
}

extern "C" {

int blp_rank_all(int model, const float* table, int64_t N, int D, int64_t ld, const float* q_fixed,
                 const float* q_rel, const int64_t* true_row, const float* q_true,
                 int64_t q_head, int64_t q_tail, const blp_filter* filter, int32_t* counts, void* workspace,
                 size_t workspace_bytes, int device, void* stream) {
    blp::QRows truth;  // stays empty (-> BAD_ARG below, after the model / size checks) unless exactly one form is given
    if ((true_row == nullptr) != (q_true == nullptr))
        truth = true_row ? blp::QRows::rows_of(table, true_row, ld) : blp::QRows::dense(q_true, D);
    return rank_all_checked(model, table, N, D, ld, blp::QRows::dense(q_fixed, D), blp::QRows::dense(q_rel, D), truth,
                            q_head, q_tail, filter, counts, workspace, workspace_bytes, device, stream);
}

int blp_rank_all_shard(int model, const float* table, int64_t N, int D, int64_t ld, const float* source, int64_t S,
                       int64_t ld_src, const int64_t* fixed_row, const float* rel_emb, int64_t R, const int64_t* rel_id,
                       const int64_t* true_row, int64_t q_head, int64_t q_tail, const blp_filter* filter, int32_t* counts,
                       void* workspace, size_t workspace_bytes, int device, void* stream) {
    if (q_head + q_tail > 0 && (!source || !fixed_row || !rel_id || !rel_emb || !true_row || R <= 0 || S <= 0))
        return fail(BLP_ERR_BAD_ARG, "blp_rank_all_shard: NULL source / fixed_row / rel_id / rel_emb / true_row, or R <= 0 / S <= 0");
    if (ld_src < D || (ld_src & 3) || !aligned16(source))
        return fail(BLP_ERR_BAD_ARG, "blp_rank_all_shard: source must be 16-byte aligned with ld_src %% 4 == 0, ld_src >= D");
    return rank_all_checked(model, table, N, D, ld, blp::QRows::rows_of(source, fixed_row, ld_src),
                            blp::QRows::rows_of(rel_emb, rel_id, D),
                            blp::QRows::rows_of(source, true_row, ld_src), q_head, q_tail, filter, counts, workspace,
                            workspace_bytes, device, stream);
}

static int64_t batches_f32_passes_per_launch(int model, int64_t N, int D, int64_t ld, int64_t n_triples, int64_t batch,
                                             int64_t block_triples) {
    if (!valid_model(model) || D <= 0 || N < 0 || n_triples < 0 || batch <= 0 || block_triples < 0) return 0;
    return blp::rank_all_batches_passes_per_launch(model, D, N, ld, n_triples, batch, block_triples);
}

static size_t batches_f32_workspace_bytes(int model, int64_t N, int D, int64_t n_triples, int64_t batch, int64_t block_triples) {
    if (!valid_model(model) || D <= 0 || N < 0 || n_triples < 0 || batch <= 0 || block_triples < 0) return 0;
    return blp::rank_all_batches_workspace_bytes(model, D, N, n_triples, batch, block_triples);
}

static int rank_all_batches_f32(int model, const float* table, int64_t N, int D, int64_t ld, const float* source, int64_t S,
                         int64_t ld_src, const int64_t* fixed_row, const float* rel_emb, int64_t R, const int64_t* rel_id,
                         const int64_t* true_row, int64_t n_triples, int64_t batch, int64_t block_triples, const blp_filter* filter,
                         int32_t* counts, void* workspace, size_t workspace_bytes, int device, void* stream) {
    if (!valid_model(model)) return fail(BLP_ERR_BAD_ARG, "blp_rank_all_batches: unknown model %d", model);
    if (n_triples < 0 || batch <= 0 || block_triples < 0 || N < 0 || ld < D)
        return fail(BLP_ERR_BAD_ARG, "blp_rank_all_batches: bad sizes (n_triples=%lld batch=%lld N=%lld)", (long long)n_triples,
                    (long long)batch, (long long)N);
    if (n_triples == 0) return BLP_OK;
    if (n_triples <= batch)  // one batch: the plain shard call (its own argument checks)
        return blp_rank_all_shard(model, table, N, D, ld, source, S, ld_src, fixed_row, rel_emb, R, rel_id, true_row, n_triples,
                                  n_triples, filter, counts, workspace, workspace_bytes, device, stream);
    if (!blp_rank_all_supported(model, D, 1, 1))
        return fail(BLP_ERR_UNSUPPORTED_DIM, "blp_rank_all_batches: D = %d not supported (see blp_rank_all_supported)", D);
    if (n_triples > (1ll << 40) || N >= (1ll << 31)) return fail(BLP_ERR_BAD_ARG, "blp_rank_all_batches: n_triples > 2^40 or N >= 2^31");
    // a ranking pass holds 2 x block_triples queries: the same 2^30 ceiling blp_rank_all puts on one block
    if (block_triples > (1ll << 29) || (block_triples == 0 && batch > (1ll << 29)))
        return fail(BLP_ERR_BAD_ARG, "blp_rank_all_batches: block_triples / batch > 2^29 (a pass ranks 2 x block_triples <= 2^30 queries)");
    if (!source || !fixed_row || !rel_id || !rel_emb || !true_row || !counts || R <= 0 || S <= 0 || (N > 0 && !table))
        return fail(BLP_ERR_BAD_ARG, "blp_rank_all_batches: NULL pointer, or R <= 0 / S <= 0");
    if (!aligned16(table) || !aligned16(source) || !aligned16(rel_emb) || !aligned16(counts) || (ld & 3) || (ld_src & 3) || ld_src < D || (D & 3))
        return fail(BLP_ERR_BAD_ARG, "blp_rank_all_batches: table / source / rel_emb / counts must be 16-byte aligned, ld %% 4 == 0, D %% 4 == 0");
    blp::FilterSpec spec;
    if (filter) {
        if (!filter->seg_lo || !filter->seg_hi || !filter->values || (filter->ent2idx && filter->ent2idx_len < 0))
            return fail(BLP_ERR_BAD_ARG, "blp_rank_all_batches: filter needs seg_lo, seg_hi and values");
        spec.lo = filter->seg_lo; spec.hi = filter->seg_hi; spec.val = filter->values; spec.exclude = filter->exclude;
        spec.ent2idx = filter->ent2idx; spec.ent2idx_len = filter->ent2idx ? filter->ent2idx_len : 0;
        spec.row_base = filter->row_base;
    }
    const size_t need = blp::rank_all_batches_workspace_bytes(model, D, N, n_triples, batch, block_triples);
    if (!workspace || workspace_bytes < need || (reinterpret_cast<uintptr_t>(workspace) & 255u))
        return fail(BLP_ERR_WORKSPACE, "blp_rank_all_batches: workspace must be 256-byte aligned and >= %zu bytes (got %zu)", need,
                    workspace_bytes);
    DeviceGuard guard(device);
    if (guard.error() != hipSuccess) return hip_fail(guard.error(), "hipSetDevice");
    int cu = 0;
    if (int rc = compute_units(device, &cu)) return rc;
    hipEvent_t ev0 = g_prof_start, ev1 = g_prof_stop;  // blp_profile_next_rank_kernel: the first ranking pass of this call
    g_prof_start = g_prof_stop = nullptr;
    hipError_t err = blp::launch_rank_all_batches(model, D, table, N, ld, source, ld_src, fixed_row, rel_emb, rel_id, true_row, n_triples,
                                                  batch, block_triples, spec, counts, workspace, cu, static_cast<hipStream_t>(stream),
                                                  ev0, ev1);
    if (err != hipSuccess) return hip_fail(err, "blp_rank_all_batches launch");
    return BLP_OK;
}

size_t blp_rank_all_batches_workspace_bytes(int model, int table_dtype, int64_t N, int D, int64_t ld, int64_t n_triples, int64_t batch,
                                              int64_t block_triples) {
    if (table_dtype == BLP_DTYPE_F32) return batches_f32_workspace_bytes(model, N, D, n_triples, batch, block_triples);
    if (!valid_model(model) || !valid_table_dtype(table_dtype) || D <= 0 || N < 0 || n_triples < 0 || batch <= 0 || block_triples < 0) return 0;
    return blp::rank_all_batches16_workspace_bytes(model, D, N, ld, n_triples, batch, block_triples);
}

int blp_rank_all_batches_native16(int model, int table_dtype, int64_t N, int D, int64_t ld, int64_t n_triples, int64_t batch,
                                  int64_t block_triples) {
    if (table_dtype == BLP_DTYPE_F32 || !valid_model(model) || !valid_table_dtype(table_dtype) || D <= 0 || N < 0 || n_triples < 0 ||
        batch <= 0 || block_triples < 0)
        return 0;
    return blp::rank_all_batches_native16(model, D, N, ld, n_triples, batch, block_triples) ? 1 : 0;
}

int64_t blp_rank_all_batches_passes_per_launch(int model, int table_dtype, int64_t N, int D, int64_t ld, int64_t n_triples, int64_t batch,
                                                 int64_t block_triples) {
    if (table_dtype == BLP_DTYPE_F32) return batches_f32_passes_per_launch(model, N, D, ld, n_triples, batch, block_triples);
    if (!valid_model(model) || !valid_table_dtype(table_dtype) || D <= 0 || N < 0 || n_triples < 0 || batch <= 0 || block_triples < 0) return 0;
    if (blp::rank_all_batches_native16(model, D, N, ld, n_triples, batch, block_triples)) return (n_triples + batch - 1) / batch;
    return n_triples <= batch ? 1 : blp::rank_all_batches_passes_per_launch(model, D, N, D, n_triples, batch, block_triples);
}

int blp_rank_all_batches(int model, const void* table, int table_dtype, int64_t N, int D, int64_t ld, const float* source, int64_t S,
                           int64_t ld_src, const int64_t* fixed_row, const float* rel_emb, int64_t R, const int64_t* rel_id,
                           const int64_t* true_row, int64_t n_triples, int64_t batch, int64_t block_triples, const blp_filter* filter,
                           int32_t* counts, void* workspace, size_t workspace_bytes, int device, void* stream) {
    if (table_dtype == BLP_DTYPE_F32)
        return rank_all_batches_f32(model, static_cast<const float*>(table), N, D, ld, source, S, ld_src, fixed_row, rel_emb, R, rel_id,
                                    true_row, n_triples, batch, block_triples, filter, counts, workspace, workspace_bytes, device, stream);
    if (!valid_model(model)) return fail(BLP_ERR_BAD_ARG, "blp_rank_all_batches: unknown model %d", model);
    if (!valid_table_dtype(table_dtype)) return fail(BLP_ERR_BAD_ARG, "blp_rank_all_batches: unknown table dtype %d", table_dtype);
    if (n_triples < 0 || batch <= 0 || block_triples < 0 || N < 0 || ld < D)
        return fail(BLP_ERR_BAD_ARG, "blp_rank_all_batches: bad sizes (n_triples=%lld batch=%lld N=%lld)", (long long)n_triples,
                    (long long)batch, (long long)N);
    if (n_triples == 0) return BLP_OK;
    const int64_t per_block = n_triples <= batch ? n_triples : 1;
    if (!blp_rank_all_supported(model, D, per_block, per_block))
        return fail(BLP_ERR_UNSUPPORTED_DIM, "blp_rank_all_batches: D = %d not supported (see blp_rank_all_supported)", D);
    if (n_triples > (1ll << 40) || N >= (1ll << 31)) return fail(BLP_ERR_BAD_ARG, "blp_rank_all_batches: n_triples > 2^40 or N >= 2^31");
    if (block_triples > (1ll << 29) || (block_triples == 0 && batch > (1ll << 29)) || (n_triples <= batch && n_triples > (1ll << 29)))
        return fail(BLP_ERR_BAD_ARG, "blp_rank_all_batches: block_triples / batch > 2^29 (a pass ranks 2 x block_triples <= 2^30 queries)");
    if (!source || !fixed_row || !rel_id || !rel_emb || !true_row || !counts || R <= 0 || S <= 0 || (N > 0 && !table))
        return fail(BLP_ERR_BAD_ARG, "blp_rank_all_batches: NULL pointer, or R <= 0 / S <= 0");
    if (!table_rows_aligned(table, table_dtype, ld) || !aligned16(source) || !aligned16(rel_emb) || !aligned16(counts) || (ld_src & 3) ||
        ld_src < D || (D & 3))
        return fail(BLP_ERR_BAD_ARG, "blp_rank_all_batches: table rows (a 16-bit table: ld %% 8 == 0) / source / rel_emb / counts must be "
                                     "16-byte aligned, D %% 4 == 0");
    blp::FilterSpec spec;
    if (filter) {
        if (!filter->seg_lo || !filter->seg_hi || !filter->values || (filter->ent2idx && filter->ent2idx_len < 0))
            return fail(BLP_ERR_BAD_ARG, "blp_rank_all_batches: filter needs seg_lo, seg_hi and values");
        spec.lo = filter->seg_lo; spec.hi = filter->seg_hi; spec.val = filter->values; spec.exclude = filter->exclude;
        spec.ent2idx = filter->ent2idx; spec.ent2idx_len = filter->ent2idx ? filter->ent2idx_len : 0;
        spec.row_base = filter->row_base;
    }
    const size_t need = blp::rank_all_batches16_workspace_bytes(model, D, N, ld, n_triples, batch, block_triples);
    if (!workspace || workspace_bytes < need || (reinterpret_cast<uintptr_t>(workspace) & 255u))
        return fail(BLP_ERR_WORKSPACE, "blp_rank_all_batches: workspace must be 256-byte aligned and >= %zu bytes (got %zu)", need,
                    workspace_bytes);
    DeviceGuard guard(device);
    if (guard.error() != hipSuccess) return hip_fail(guard.error(), "hipSetDevice");
    int cu = 0;
    if (int rc = compute_units(device, &cu)) return rc;
    hipEvent_t ev0 = g_prof_start, ev1 = g_prof_stop;  // blp_profile_next_rank_kernel: the first ranking pass of this call
    g_prof_start = g_prof_stop = nullptr;
    hipError_t err = blp::launch_rank_all_batches16(model, D, table, table_dtype, N, ld, source, ld_src, fixed_row, rel_emb, rel_id, true_row,
                                                    n_triples, batch, block_triples, spec, counts, workspace, cu,
                                                    static_cast<hipStream_t>(stream), ev0, ev1);
    if (err != hipSuccess) return hip_fail(err, "blp_rank_all_batches launch");
    return BLP_OK;
}

int blp_profile_next_rank_kernel(void* start_event, void* stop_event) {
    if ((start_event == nullptr) != (stop_event == nullptr))
        return fail(BLP_ERR_BAD_ARG, "blp_profile_next_rank_kernel: give both events or neither");
    g_prof_start = static_cast<hipEvent_t>(start_event);
    g_prof_stop = static_cast<hipEvent_t>(stop_event);
    return BLP_OK;
}

int blp_rank_from_scores(const float* scores, int64_t Q, int64_t N, int64_t ld, const int64_t* true_idx,
                         const float* true_score, const int64_t* filt_rowptr, const int64_t* filt_col,
                         int32_t* counts, int device, void* stream) {
    if (Q < 0 || N < 0 || ld < N) return fail(BLP_ERR_BAD_ARG, "blp_rank_from_scores: negative size or ld < N");
    if (Q == 0) return BLP_OK;
    if (!scores || !counts) return fail(BLP_ERR_BAD_ARG, "blp_rank_from_scores: NULL scores / counts");
    if ((true_idx == nullptr) == (true_score == nullptr))
        return fail(BLP_ERR_BAD_ARG, "blp_rank_from_scores: exactly one of true_idx / true_score must be given");
    if ((filt_rowptr == nullptr) != (filt_col == nullptr))
        return fail(BLP_ERR_BAD_ARG, "blp_rank_from_scores: filt_rowptr and filt_col go together");
    if (!aligned16(counts)) return fail(BLP_ERR_BAD_ARG, "blp_rank_from_scores: counts must be 16-byte aligned");
    DeviceGuard guard(device);
    if (guard.error() != hipSuccess) return hip_fail(guard.error(), "hipSetDevice");
    hipError_t err = blp::launch_rank_from_scores(scores, Q, N, ld, true_idx, true_score, filt_rowptr, filt_col, counts,
                                                  static_cast<hipStream_t>(stream));
    if (err != hipSuccess) return hip_fail(err, "blp_rank_from_scores launch");
    return BLP_OK;
}

int blp_rank_metrics(const int32_t* counts, int64_t Q, const int32_t k_values[3], float* rr, uint8_t* hits,
                     int device, void* stream) {
    if (Q < 0 || (Q > 0 && (!counts || !rr || !hits || !k_values)))
        return fail(BLP_ERR_BAD_ARG, "blp_rank_metrics: NULL argument or negative Q");
    if (!aligned16(counts)) return fail(BLP_ERR_BAD_ARG, "blp_rank_metrics: counts must be 16-byte aligned");
    DeviceGuard guard(device);
    if (guard.error() != hipSuccess) return hip_fail(guard.error(), "hipSetDevice");
    hipError_t err = blp::launch_rank_metrics(counts, Q, k_values, rr, hits, static_cast<hipStream_t>(stream));
    if (err != hipSuccess) return hip_fail(err, "blp_rank_metrics launch");
    return BLP_OK;
}

int blp_rank_metric_sums(const int32_t* counts, int64_t Q, const int32_t k_values[3], double* sums, int device,
                         void* stream) {
    if (Q < 0 || !k_values || !sums || (Q > 0 && !counts))
        return fail(BLP_ERR_BAD_ARG, "blp_rank_metric_sums: NULL pointer or negative Q");
    if (!aligned16(counts)) return fail(BLP_ERR_BAD_ARG, "blp_rank_metric_sums: counts must be 16-byte aligned");
    DeviceGuard guard(device);
    if (guard.error() != hipSuccess) return hip_fail(guard.error(), "hipSetDevice");
    hipError_t err = blp::launch_rank_metric_sums(counts, Q, k_values, sums, static_cast<hipStream_t>(stream));
    if (err != hipSuccess) return hip_fail(err, "blp_rank_metric_sums launch");
    return BLP_OK;
}

int blp_score_fwd(int model, int D, int64_t M0, int64_t M1, const float* heads, int64_t h_s0, int64_t h_s1,
                  const float* tails, int64_t t_s0, int64_t t_s1, const float* rels, int64_t r_s0, int64_t r_s1,
                  float* out, int device, void* stream) {
    if (!valid_model(model)) return fail(BLP_ERR_BAD_ARG, "blp_score_fwd: unknown model %d", model);
    if (int rc = check_score_dim(model, D)) return rc;
    if (M0 < 0 || M1 < 0) return fail(BLP_ERR_BAD_ARG, "blp_score_fwd: negative output shape");
    if (M0 * M1 == 0) return BLP_OK;
    if (!heads || !tails || !rels || !out) return fail(BLP_ERR_BAD_ARG, "blp_score_fwd: NULL pointer");
    DeviceGuard guard(device);
    if (guard.error() != hipSuccess) return hip_fail(guard.error(), "hipSetDevice");
    hipError_t err = blp::launch_score_fwd(model, D, M0, M1, rows(heads, h_s0, h_s1), rows(tails, t_s0, t_s1),
                                           rows(rels, r_s0, r_s1), out, static_cast<hipStream_t>(stream));
    if (err != hipSuccess) return hip_fail(err, "blp_score_fwd launch");
    return BLP_OK;
}

int blp_score_bwd(int model, int D, int64_t M0, int64_t M1, const float* heads, int64_t h_s0, int64_t h_s1,
                  const float* tails, int64_t t_s0, int64_t t_s1, const float* rels, int64_t r_s0, int64_t r_s1,
                  const float* grad_out, float* grad_heads, float* grad_tails, float* grad_rels, int device,
                  void* stream) {
    if (!valid_model(model)) return fail(BLP_ERR_BAD_ARG, "blp_score_bwd: unknown model %d", model);
    if (int rc = check_score_dim(model, D)) return rc;
    if (M0 < 0 || M1 < 0) return fail(BLP_ERR_BAD_ARG, "blp_score_bwd: negative output shape");
    if (M0 * M1 == 0) return BLP_OK;
    if (!heads || !tails || !rels || !grad_out) return fail(BLP_ERR_BAD_ARG, "blp_score_bwd: NULL pointer");
    DeviceGuard guard(device);
    if (guard.error() != hipSuccess) return hip_fail(guard.error(), "hipSetDevice");
    hipError_t err = blp::launch_score_bwd(model, D, M0, M1, rows(heads, h_s0, h_s1), rows(tails, t_s0, t_s1),
                                           rows(rels, r_s0, r_s1), grad_out, grad_heads, grad_tails, grad_rels,
                                           static_cast<hipStream_t>(stream));
    if (err != hipSuccess) return hip_fail(err, "blp_score_bwd launch");
    return BLP_OK;
}

static int check_inbatch(const char* who, int model, int loss, const void* ent, const void* rel, const void* neg_idx,
                         int B, int K, int D) {
    if (!valid_model(model)) return fail(BLP_ERR_BAD_ARG, "%s: unknown model %d", who, model);
    if (loss != BLP_LOSS_MARGIN && loss != BLP_LOSS_NLL) return fail(BLP_ERR_BAD_ARG, "%s: unknown loss %d", who, loss);
    if (B <= 0 || K <= 0) return fail(BLP_ERR_BAD_ARG, "%s: B and K must be positive (B=%d K=%d)", who, B, K);
    if (int rc = check_score_dim(model, D)) return rc;
    if (!ent || !rel || !neg_idx) return fail(BLP_ERR_BAD_ARG, "%s: NULL pointer", who);
    return BLP_OK;
}

static int check_dtypes(const char* who, int ent_dtype, int rel_dtype) {
    const bool known = ent_dtype >= BLP_DTYPE_F32 && ent_dtype <= BLP_DTYPE_BF16;
    if (!known || (rel_dtype != ent_dtype && rel_dtype != BLP_DTYPE_F32))
        return fail(BLP_ERR_BAD_ARG, "%s: ent_dtype %d / rel_dtype %d (rel must have ent's type or be f32)", who,
                    ent_dtype, rel_dtype);
    return BLP_OK;
}

int blp_inbatch_loss_fwd(int model, int loss, int ent_dtype, int rel_dtype, const void* ent_embs,
                           const void* rel_vecs, const int64_t* neg_idx, int B, int K, int D, float regularizer,
                           float* out_loss, float* save_pos, float* save_neg, int32_t* ticket, int device, void* stream) {
    if (int rc = check_inbatch("blp_inbatch_loss_fwd", model, loss, ent_embs, rel_vecs, neg_idx, B, K, D)) return rc;
    if (int rc = check_dtypes("blp_inbatch_loss_fwd", ent_dtype, rel_dtype)) return rc;
    if (!out_loss || !save_pos || !save_neg || !ticket) return fail(BLP_ERR_BAD_ARG, "blp_inbatch_loss_fwd: NULL output / ticket");
    if (reinterpret_cast<uintptr_t>(save_pos) & 7) return fail(BLP_ERR_BAD_ARG, "blp_inbatch_loss_fwd: save_pos must be 8-byte aligned");
    DeviceGuard guard(device);
    if (guard.error() != hipSuccess) return hip_fail(guard.error(), "hipSetDevice");
    hipError_t err = blp::launch_inbatch_loss_fwd(model, loss, ent_dtype, rel_dtype, ent_embs, rel_vecs, neg_idx, B, K, D,
                                                  regularizer, out_loss, save_pos, save_neg, reinterpret_cast<unsigned*>(ticket),
                                                  static_cast<hipStream_t>(stream));
    if (err != hipSuccess) return hip_fail(err, "blp_inbatch_loss_fwd launch");
    return BLP_OK;
}

int blp_inbatch_loss_bwd(int model, int loss, int ent_dtype, int rel_dtype, const void* ent_embs,
                           const void* rel_vecs, const int64_t* neg_idx, int B, int K, int D, float regularizer,
                           const float* grad_loss, const float* save_pos, const float* save_neg, void* grad_ent,
                           void* grad_rel, int device, void* stream) {
    if (int rc = check_inbatch("blp_inbatch_loss_bwd", model, loss, ent_embs, rel_vecs, neg_idx, B, K, D)) return rc;
    if (int rc = check_dtypes("blp_inbatch_loss_bwd", ent_dtype, rel_dtype)) return rc;
    if (!grad_loss || !save_pos || !save_neg || !grad_ent || !grad_rel)
        return fail(BLP_ERR_BAD_ARG, "blp_inbatch_loss_bwd: NULL pointer");
    DeviceGuard guard(device);
    if (guard.error() != hipSuccess) return hip_fail(guard.error(), "hipSetDevice");
    hipError_t err = blp::launch_inbatch_loss_bwd(model, loss, ent_dtype, rel_dtype, ent_embs, rel_vecs, neg_idx, B, K, D,
                                                  regularizer, grad_loss, save_pos, save_neg, grad_ent, grad_rel,
                                                  static_cast<hipStream_t>(stream));
    if (err != hipSuccess) return hip_fail(err, "blp_inbatch_loss_bwd launch");
    return BLP_OK;
}

int blp_inbatch_loss_fwd_launches(int model, int B, int K, int D, float regularizer) {
    if (!valid_model(model) || B <= 0 || K <= 0 || D <= 0) return 0;
    return blp::inbatch_loss_fwd_launches(model, B, K, D, regularizer > 0.0f);
}

size_t blp_inbatch_loss_save_floats(int model, int B, int K, int D) {
    if (!valid_model(model) || B <= 0 || K <= 0 || D <= 0) return 0;
    return blp::inbatch_loss_save_floats(model, B, K, D);
}

}  // extern "C"
