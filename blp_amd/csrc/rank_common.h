// rank_common.h -- helpers shared by the ranking kernels (rank_all.hip: exact one-lane-per-candidate VALU
// kernels; rank_sad*.hip, rank_gemm.hip: the pre-pass paths).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <utility>

#include "launch.h"
#include "score_core.h"

namespace blp {

typedef __attribute__((address_space(1))) const void* global_cptr;
typedef __attribute__((address_space(3))) void* lds_ptr;

__device__ __forceinline__ void wave_lds_sync() {
    // LDS operations of one wave execute in issue order; this only stops the compiler from
    // moving LDS accesses across the hand-off point.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int D>
__device__ __forceinline__ void load_row(float (&e)[D], const float* __restrict__ row) {
#pragma unroll
    for (int d = 0; d < D; d += 4) {
        const float4 v = *reinterpret_cast<const float4*>(row + d);
        e[d] = v.x; e[d + 1] = v.y; e[d + 2] = v.z; e[d + 3] = v.w;
    }
}

// Hand-issued scalar loads (wave-uniform data straight into SGPRs).  Scalar loads return out of order,
// so every wait on them is a full drain: callers request the next 64-byte chunk before a block of VALU
// work and drain after it.  stouch<> reads one dword per 64-byte line (result discarded) to pull lines
// into the scalar cache ahead of use.
typedef float sf16 __attribute__((ext_vector_type(16)));

template <int OFF>
__device__ __forceinline__ sf16 sload16(const float* base) {
    sf16 v;
    asm volatile("s_load_dwordx16 %0, %1, %2" : "=s"(v) : "s"(base), "i"(OFF) : "memory");
    return v;
}
__device__ __forceinline__ void sdrain(sf16& a) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a) : : "memory"); }
__device__ __forceinline__ void sdrain(sf16& a, sf16& b) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a), "+s"(b) : : "memory"); }

template <int OFF>
__device__ __forceinline__ void stouch_line(const float* row) {
    float sink;
    asm volatile("s_load_dword %0, %1, %2" : "=s"(sink) : "s"(row), "i"(OFF) : "memory");
}
template <int... Ls>
__device__ __forceinline__ void stouch_lines(const float* row, std::integer_sequence<int, Ls...>) {
    (stouch_line<Ls * 64>(row), ...);
}
template <int BYTES>
__device__ __forceinline__ void stouch(const float* row) {  // one dword per 64-B line, result discarded
    stouch_lines(row, std::make_integer_sequence<int, BYTES / 64>{});
}

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// Row of `table` that entry k of query q's filter list removes, or -1 if it removes nothing here.
__device__ __forceinline__ int64_t filter_row(const FilterSpec& f, int64_t q, int64_t k, int64_t N) {
    int64_t v = f.val[k];
    if (f.exclude && v == f.exclude[q]) return -1;
    if (f.ent2idx) v = (uint64_t)v < (uint64_t)f.ent2idx_len ? f.ent2idx[v] : -1;
    v -= f.row_base;
    return (uint64_t)v < (uint64_t)N ? v : -1;
}

// Last step of every ranking path: counts[q] = {gt, ge, gt - fgt, ge - fge} from the packed 64-bit accumulators
// (query q's is the sum of acc[p * Q + q] over p < n_partials: the small-block kernel leaves one partial per tile slot),
// fgt / fge = the filtered candidates that score above / at least the true entity (one wave per query, exact
// Scorer<> arithmetic).  Without a filter the last two equal the first two.
hipError_t launch_filter_finalize(int model, int D, const float* table, int64_t N, int64_t ld, const QRows q_fixed,
                                  const QRows q_rel, const float* key_true, int64_t q_head, int64_t q_tail,
                                  const FilterSpec& filter, const unsigned long long* acc, int32_t* counts,
                                  hipStream_t stream, int n_partials = 1);

// ---- the bounded worst case of the pre-pass paths (round 5) --------------------------------------------------------------
// A pre-pass pays off while it leaves little to the exact path.  When a device-side counter says it has not -- the pair lists
// ran (nearly) full: exact ties with the true entity on whole percents of the table -- the refinement kernels stand down and
// the EXACT kernel of rank_all.hip re-ranks the block from scratch, so that a block never costs more than pre-pass + exact
// kernel (TransE 3 + 8 ms, bilinear 1 + 15 ms for the FB15k-237 block; the refinement alone took 130 / 70 ms on tables with 5 %
// duplicates of the true entity).  Everything is decided ON THE DEVICE: every kernel involved reads the counter and returns
// at once when it is not its turn (`gate`: the counter; heavy <=> *gate >= gate_min).  gate == nullptr: no gating.
struct Gate { const unsigned* counter; unsigned heavy_from; };
__device__ __forceinline__ bool gate_heavy(const Gate& g) { return g.counter != nullptr && *g.counter >= g.heavy_from; }
// What a refinement kernel does INSTEAD of refining when the gate says heavy (one launch fewer than a kernel of its own would
// be: ~5 us of every common-case block): zero the counts the pre-pass took and materialise the coefficient rows of the exact
// kernel, grid-strided over whatever grid the caller has.  coef: exact_fallback_coef_floats(...) floats.
struct FallbackPrep { float* coef; unsigned long long* acc; };
// floats of coefficient rows the exact re-ranking needs for (q_head + q_tail) queries; blocks below kFallbackMinPairs (or ranked
// in several candidate slabs) have no fallback: 0
constexpr int64_t kFallbackMinPairs = (int64_t)1 << 27;
size_t exact_fallback_coef_floats(int D, int64_t q_head, int64_t q_tail);
// if heavy: rank every query against every row with rank_tiles_kernel (counts into acc, which the last refinement kernel of the
// path zeroed together with writing the coefficient rows: fallback_prep below; key_true as computed by the pre-pass path);
// else: one launch that returns at once
hipError_t launch_exact_fallback(int model, int D, const float* table, int64_t N, int64_t ld, const QRows q_fixed, const QRows q_rel,
                                 int64_t q_head, int64_t q_tail, float* coef, const float* key_true, unsigned long long* acc, Gate gate,
                                 int n_cu, hipStream_t stream);
// where the tail-side coefficient rows start inside `coef`
__host__ __device__ inline size_t fallback_coef_tail_offset(int D, int64_t q_head) { return (((size_t)q_head * 2 * D + 63) / 64) * 64; }

constexpr int64_t kTrueKeyLaneMaxQueries = 2048;  // up to here true keys are one lane per query (exact_coop.h: true_key_lane)

// rank_all.hip: true-entity keys by the exact routine (and the Q rank-count accumulators `acc` zeroed on the way)
hipError_t launch_true_keys(int model, int D, const QRows q_fixed, const QRows q_rel, int64_t q_head, int64_t q_tail,
                            const QRows q_true,
                            float* key_true, unsigned long long* acc, hipStream_t stream);

// rank_small.hip: the exact f32 kernel for small blocks (coefficients computed in the kernel, LDS broadcasts, TransE
// chains interleaved): the ranking pass only, between launch_true_keys and launch_filter_finalize.
constexpr long long kSmallMaxPairsTransE = 4000000;    // above (== kSadMinPairs): the fixed-point pre-pass
constexpr long long kSmallMaxPairsTransESgpr = 6000000;  // the same for tables of up to kSmallMaxSlots tiles (tools/exact_small_probe.py: 440 queries x 14 541 rows; the scalar-register kernel: up to kSmallSgprMaxTiles tiles)
constexpr long long kSmallMaxPairsBilinear = 400000;   // above (and from 32 queries on): the bf16 x 3 MFMA pre-pass (tools/exact_small_probe.py)
constexpr int kSmallMaxSlots = 256;                    // partial counts per query it leaves (one per tile slot)
constexpr int kSmallSgprMaxTiles = 1024;               // tables the scalar-register TransE kernel takes (tiles beyond the slots share them)
constexpr long long kSmallMaxQueries = 4096;           // whatever the knob says
bool rank_small_applicable(int model, int D, int64_t N, int64_t q_head, int64_t q_tail);
int rank_small_slots(int64_t N);  // partial[slot * Q + q], slot < rank_small_slots(N): what the kernel writes (every entry)
bool rank_small_wants_coef(int model, int D, int64_t N);  // its kernel reads materialised coefficient rows (coef_head / coef_tail)
hipError_t launch_rank_small(int model, int D, const float* table, int64_t N, int64_t ld, const QRows q_fixed,
                             const QRows q_rel, const float* coef_head, const float* coef_tail, const float* key_true,
                             int64_t q_head, int64_t q_tail, unsigned long long* partial, int n_cu, hipStream_t stream);

// rank_stream.hip: <= 4 + 4 queries, long table: the table streamed through a load pipeline that never drains (TransE: a
// wave's ring of 32-column pieces; bilinear models: a workgroup's double-buffered LDS tile).  The ranking pass only:
// coefficients and true keys as for rank_tiles<STATIC>; adds to acc[q].
bool rank_stream_applicable(int model, int D, int64_t N, int64_t ld, int64_t q_head, int64_t q_tail);
// The bilinear models' approximate keys (rank_stream.hip, DOT): per pass kStreamDotRows operand rows W_q of D floats --
// query q of the pass (heads first) in row stream_dot_row(q): the workgroup-tile kernel's wave w has queries w and w + 4
// in neighbouring rows -- and two band factors per query, band[2 q] = C u ||B_q|| (times the row's norm) and
// band[2 q + 1] = the absolute band against rows whose squares underflow; written by rank_all.hip's preparation launch
// (dot_prepare).  wq == nullptr: order-exact keys only.  The ring kernel re-scores undecided pairs from the queries'
// own vectors: query q of the pass is q0 + q of q_fixed / q_rel.
constexpr int kStreamDotRows = 8;
__host__ __device__ constexpr int stream_dot_row(int q) { return (q % 4) * 2 + q / 4; }
struct StreamDot {
    const float* wq = nullptr;
    const float* band = nullptr;
    QRows q_fixed, q_rel;
    int64_t q0 = 0;
};
bool rank_stream_wants_dot(int model, int D, int64_t N, int64_t ld, int64_t q_head, int64_t q_tail, bool passes = false);
// Many passes of `batch` triples each (the last: what is left of n) in ONE launch of a ring kernel, in the layout
// rank_all.hip's prep_passes_kernel leaves: pass p's queries are 2 p batch .. of the call's key_true / acc / band arrays,
// its coefficient rows start at (p batch) x (floats per triple), its operand rows at p x kStreamDotRows x D.  acc_slots
// replicas of the accumulators (slot s of query q: acc[s * 2 n + q]).  n_passes <= 1: the single pass the pointers describe.
struct StreamPasses {
    int n_passes = 1;
    int batch = 0;
    int64_t n = 0;
    int acc_slots = 1;
};
constexpr int kStreamAccSlots = 16;
bool rank_stream_takes_passes(int model, int D, int64_t N, int64_t ld, int64_t batch, int64_t n);  // all passes in one launch?
hipError_t launch_rank_stream(int model, int D, const float* table, int64_t N, int64_t ld, const float* coef_head,
                              const float* coef_tail, const float* key_true, int64_t q_head, int64_t q_tail,
                              unsigned long long* acc, const StreamDot& dot, int n_cu, hipStream_t stream,
                              const StreamPasses& passes = StreamPasses());

// rank_stream16.hip: the ring kernels over a 16-bit candidate table (table_elem.h: dtype), every pass of a call in one launch
// (n_passes >= 1); the bilinear models with approximate keys (dot.wq) always.  D = 128 or 256.
bool rank_stream16_takes_passes(int model, int D, int64_t N, int64_t ld, int64_t batch, int64_t n);
hipError_t launch_rank_stream16(int model, int D, int dtype, const void* table, int64_t N, int64_t ld, const float* coef,
                                const float* key_true, int64_t q_head, int64_t q_tail, unsigned long long* acc,
                                const StreamDot& dot, int n_cu, hipStream_t stream, const StreamPasses& passes);

// Measurement (include/blp_hip.h: blp_rank_all_prepass_stats): what the pre-pass of the last launch_rank_all on `workspace`
// left to the exact path.  path: 0 = no pre-pass took this block (exact kernels), 1 = TransE v_sad_u16, 2 = bf16 x 3 MFMA,
// 3 = f32-chain MFMA (not counted: listed = flagged_rows = -1), 4 = any-width TransE (not counted).  Last candidate slab only.
struct PrepassStats { long long pairs, listed, flagged_rows; int path; };
hipError_t prepass_stats(int model, int D, int64_t N, int64_t q_head, int64_t q_tail, const void* workspace, PrepassStats* out,
                         hipStream_t stream);
hipError_t gemm_prepass_stats(int D, int64_t N, int64_t q_head, int64_t q_tail, const void* workspace, int device, PrepassStats* out, hipStream_t stream);
hipError_t sad_prepass_stats(int D, int64_t N, int64_t q_head, int64_t q_tail, const void* workspace, PrepassStats* out, hipStream_t stream);
// one workgroup-strided pass: out[0] += sum popc(flags[i]); out[1] += (mask_entries ? sum popc(pairs[i].y & 0xffff) : 0) over *n_pairs entries
hipError_t launch_count_bits(const unsigned* flags, int64_t n_words, const uint2* pairs, const unsigned* n_pairs, bool mask_entries,
                             unsigned long long host_out[3], hipStream_t stream, unsigned max_entries = 0xffffffffu);

// rank_gemm.hip: the run-time guard behind the bf16 band's one empirical assumption (how v_mfma_f32_32x32x16_bf16 rounds its
// accumulation): per device 0 = not tested yet, 1 = holds, 2 = violated (bilinear blocks take the f32-chain pre-pass).
// mfma_accum_selftest runs it once per device on `stream` (scratch: >= 512 bytes of device memory) and waits for it; 0 = no
// verdict now (the stream is being captured).
int mfma_accum_state(int device);
float mfma_accum_worst(int device);  // the largest |S~ - S3| / (262 u T) the self-test saw (fails at 0.5)
int mfma_accum_selftest(int device, void* scratch, hipStream_t stream, hipError_t* err_out);
#ifdef BLP_TEST_HOOKS
void mfma_accum_reset(int device);
#endif
// rank_gemm.hip: bilinear models as an MFMA GEMM + error band + exact refinement.
void gemm_set_dump(float* s, float* eps);  // blp_debug_gemm_dump (tests)
bool rank_gemm_applicable(int model, int D, int64_t q_head, int64_t q_tail);
size_t rank_gemm_workspace_bytes(int model, int D, int64_t N, int64_t q_head, int64_t q_tail);
hipError_t launch_rank_all_gemm(int model, int D, const float* table, int64_t N, int64_t ld,
                                const QRows q_fixed, const QRows q_rel, const QRows q_true, int64_t q_head, int64_t q_tail,
                                const FilterSpec& filter, int32_t* counts,
                                void* workspace, int n_cu, hipStream_t stream, hipEvent_t ev_start,
                                hipEvent_t ev_stop);

// rank_sad.hip: TransE with many queries as a 16-bit fixed-point v_sad_u16 pre-pass + error band +
// exact refinement.
bool rank_sad_applicable(int model, int D, int64_t N, int64_t q_head, int64_t q_tail);
size_t rank_sad_workspace_bytes(int model, int D, int64_t N, int64_t q_head, int64_t q_tail);
hipError_t launch_rank_all_sad(int D, const float* table, int64_t N, int64_t ld, const QRows q_fixed,
                               const QRows q_rel, const QRows q_true, int64_t q_head,
                               int64_t q_tail, const FilterSpec& filter, int32_t* counts,
                               void* workspace, int n_cu, hipStream_t stream, hipEvent_t ev_start,
                               hipEvent_t ev_stop);

// rank_sad_wide.hip: the same pre-pass for TransE at any width (D % 4 == 0, D <= 1024; the BOW / DKRL
// encoders' 300 and 768), 128 elements of the candidate row in registers at a time.
bool rank_sad_wide_applicable(int model, int D, int64_t q_head, int64_t q_tail);
size_t rank_sad_wide_workspace_bytes(int model, int D, int64_t N, int64_t q_head, int64_t q_tail);
hipError_t launch_rank_all_sad_wide(int D, const float* table, int64_t N, int64_t ld, const QRows q_fixed,
                                    const QRows q_rel, const QRows q_true, int64_t q_head,
                                    int64_t q_tail, const FilterSpec& filter,
                                    int32_t* counts, void* workspace, int n_cu, hipStream_t stream,
                                    hipEvent_t ev_start, hipEvent_t ev_stop);

// Exact sweeps of flagged (query, candidate segment) bits: a workgroup of four waves owns kSweepQueries
// consecutive queries.  It scans their flag words together (coalesced; or takes every query when `all`, the
// fallback when the pre-pass did not run), leaves the offsets of the flagged queries, in order, in list[],
// and its waves then take them round-robin.  Flags are rare: a workgroup per four queries (one wave each) spent
// 37 us launching 26 k workgroups that found nothing (FB15k-237 block); this layout: 8 us.
constexpr int kSweepQueries = 64;
__device__ __forceinline__ int flagged_queries(const unsigned* __restrict__ flags, int64_t q_base, int64_t Q,
                                               int words_per_query, bool all, int* list, int* n_list) {
    const int nq = (int)(Q - q_base < kSweepQueries ? Q - q_base : kSweepQueries);
    if (threadIdx.x < kSweepQueries) list[threadIdx.x] = (int)threadIdx.x < nq && all;
    __syncthreads();
    if (!all) {  // the workgroup reads its queries' flag words (contiguous) together, whatever their number
        const unsigned* base = flags + q_base * words_per_query;
        const int total = nq * words_per_query;
        for (int i = threadIdx.x; i < total; i += blockDim.x)
            if (base[i]) list[i / words_per_query] = 1;  // every writer stores the same value
        __syncthreads();
    }
    if (threadIdx.x < 64) {  // first wave: marks -> ordered list of offsets (all lanes read before any writes)
        const bool any = list[threadIdx.x] != 0;
        const unsigned long long mask = __ballot(any);
        if (any) list[__popcll(mask & ((1ull << threadIdx.x) - 1ull))] = (int)threadIdx.x;
        if (threadIdx.x == 0) *n_list = __popcll(mask);
    }
    __syncthreads();
    return *n_list;
}

}  // namespace blp
