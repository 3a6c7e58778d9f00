// stream_common.h -- what the ring kernels of rank_stream.hip (an f32 table) and rank_stream16.hip (a 16-bit table) share:
// the piece loads, the view of a pass of a many-pass launch, the hand-issued scalar loads pinned among the arithmetic, the
// approximate keys' band and the on-the-spot exact re-scoring of undecided rows.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dot_band.h"
#include "exact_coop.h"
#include "rank_common.h"
#include "score_core.h"
#include "table_elem.h"
#include "tile.h"

#pragma clang fp contract(off)

namespace blp {

constexpr int kStreamQ = 4;  // queries per side (== rank_all.hip's kQB: what its static mode takes)
constexpr int64_t kStreamWgMinRows = 1700000;  // TransE: from here on the workgroup-tile kernel (launch_rank_stream)

typedef float f32x4 __attribute__((ext_vector_type(4)));

// The 8 loads of one 32-column piece: piece s of the tile whose first row is `base` (byte offsets boff[i] of this lane's
// part of rows 8i .. 8i + 7, clamped to the table by the caller).  Streamed once: non-temporal.
template <class Off>
__device__ __forceinline__ void piece_fetch(f32x4 (&b)[8], const void* __restrict__ base, const Off (&boff)[8], int s) {
#ifdef BLP_STREAM_CONTIG  // experiment (with BLP_STREAM_NULL): a piece = 8 KB contiguous -- 16 whole rows -- instead of 64 rows x 128 B
    constexpr int kPieceStep = 8192;
#else
    constexpr int kPieceStep = kSubCols * 4;
#endif
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#ifdef BLP_STREAM_PLAIN_LOADS  // experiment (tools/step_ab.py, tools/table16_probe.py): cacheable loads -- does a table that fits the Infinity Cache stay there between passes?
        b[i] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(base) + boff[i] + s * kPieceStep);
#else
        b[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(base) + boff[i] + s * kPieceStep));
#endif
    }
}

// MANY PASSES IN ONE LAUNCH (StreamPasses, rank_common.h; the reference's Wikidata5M evaluation is a pass per two triples).
// A pass launched on its own ramps up and drains: on a 1/8 shard of the 4.6 M-row table that is several of its ~45 us.
// The ring kernels take all passes of a call as ONE index space of (pass, round of four tiles), workgroup b walking
// rounds b, b + gridDim.x, ...: a wave crosses from pass p's last tile to pass p + 1's first with its ring full, and the
// load pipeline of the chip never drains between passes.  A wave that enters a new pass flushes its counts straight to
// the accumulators (replicated acc_slots times -- slot = workgroup mod acc_slots -- so that a few thousand waves do not
// queue on one address; the finalisation adds the slots) and re-derives the pass's pointers; nothing synchronises.
struct PassView {  // pass p of a call in the layout of rank_all.hip's prep_passes_kernel (or the single pass as given)
    int q_head, q_tail;
    int64_t first2;  // queries of the call before this pass (2 x its first triple)
};
__device__ __forceinline__ PassView pass_view(const StreamPasses& ps, int p, int q_head, int q_tail) {
    if (ps.n_passes <= 1) return PassView{q_head, q_tail, 0};
    const int64_t first = (int64_t)p * ps.batch, left = ps.n - first;
    const int nb = (int)(left < ps.batch ? left : ps.batch);
    return PassView{nb, nb, 2 * first};
}

// Hand-issued scalar loads whose place among the arithmetic is pinned by an operand (see the approximate-key kernel below).
template <int OFF>
__device__ __forceinline__ void sload16_pinned(sf16& v, const float* base, float& pin) {
    asm volatile("s_load_dwordx16 %0, %2, %3" : "=s"(v), "+v"(pin) : "s"(base), "i"(OFF) : "memory");
}
__device__ __forceinline__ void sdrain_pinned(sf16& a, float& pin) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a), "+v"(pin) : : "memory");
}
__device__ __forceinline__ void sdrain_pinned(sf16& a, float& pin, float& pin2) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a), "+v"(pin), "+v"(pin2) : : "memory");
}
__device__ __forceinline__ void sdrain_pinned2(sf16& a, sf16& b, float& pin) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a), "+s"(b), "+v"(pin) : : "memory");
}

// The bilinear models' approximate keys (the ring kernel further down): a key is first taken as the plain dot product
// <W_q, e> -- one chain of D fused multiply-adds against the query's GEMM operand row (dot_band.h; 128 VALU instructions
// where the reference's order costs 512 - 790) -- and decided against the exact true key within the band of rank_gemm.hip's
// f32-chain kernel, eps = C u ||B_q|| ||e|| with ||e|| from the same registers; only rows that stay undecided (the true
// entity itself, near-ties, non-finite values, magnitudes outside the band's premises) are scored by the order-exact
// routine as well and count by that key.  Counts are the exact kernels' in every case.
struct DotBand {  // one query's band against a row of squared norm ss: see dot_prepare()
    float kt, eq, et, guard;
};
__device__ __forceinline__ DotBand dot_band_of(const float* __restrict__ key_true, const float* __restrict__ band, int q) {
    DotBand b;
    b.kt = key_true[q];
    b.eq = band[2 * q];
    b.et = band[2 * q + 1];
    b.guard = fabsf(b.kt) * 2.4e-7f + 1e-35f;  // rounding of kt +- eps themselves; underflowing products
    return b;
}
// certainly above (gt) / certainly below (lt) the true key; neither: undecided.  nrow = ||e|| rounded up, tiny = the
// squares underflow (every |e_k| <= 1.0001e-15: the band is the absolute one, et); premises broken (overflowing
// magnitudes, NaN anywhere): eps = inf or NaN, both compares fail.
__device__ __forceinline__ void dot_decide(float v, const DotBand& b, float nrow, bool tiny, bool& gt, bool& lt) {
    float eps = tiny ? b.et : b.eq * nrow;
    eps = eps < 1e30f ? eps + b.guard : __builtin_inff();
    gt = v > b.kt + eps;
    lt = v < b.kt - eps;
#ifdef BLP_DOT_ALL_EXACT  // debugging aid: nothing is decided, every row takes the exact routine
    gt = lt = false;
#endif
}

// The undecided rows `und` (a lane mask) of the tile whose first row is row0, against one query (f, r: its two vectors;
// kt: its true key), by coop_score -- 32 lanes per pair, the order-exact arithmetic of every other exact path, the row
// re-read from the cache it has just passed through; two rows per trip, one per 32-lane half (the upper half without a
// row of its own repeats the lower one's).  Returns the rows at or above the true key as gt | ge << 32.
template <int MODEL, int D, class TE = float>
__device__ __forceinline__ unsigned long long exact_undecided(const TE* __restrict__ table, int64_t ld, int64_t row0,
                                                              const float* f, const float* r, bool head,
                                                              unsigned long long und, float kt, int lane) {
    const bool upper = lane >= 32;
    const int sub = lane & 31;
    unsigned gt = 0, ge = 0;
    while (und) {
        const int r0 = __builtin_ctzll(und);
        und &= und - 1;
        int r1 = -1;
        if (und) { r1 = __builtin_ctzll(und); und &= und - 1; }
        const bool mine = sub == 0 && (!upper || r1 >= 0);
        const TE* e = table + (row0 + (upper && r1 >= 0 ? r1 : r0)) * ld;
        const float key = head ? coop_score<MODEL, HEAD, D>(e, f, r, sub) : coop_score<MODEL, TAIL, D>(e, f, r, sub);
        gt += __popcll(__ballot(mine && key > kt));
        ge += __popcll(__ballot(mine && key >= kt));
    }
    return (unsigned long long)gt | ((unsigned long long)ge << 32);
}

}  // namespace blp
