// knobs.h -- test / A-B hooks (blp_debug_set_knob in include/blp_hip.h).  They exist ONLY in the test build of the library
// (-DBLP_TEST_HOOKS -> libblp_hip.hooks.so, loaded by tests/ and tools/): process-wide relaxed atomics, 0 = automatic.  In
// the production library knob() is the constant 0 -- every `if (knob(...))` folds away, nothing is exported, the library
// has no mutable process-wide state and there is no getenv() anywhere, so blp_rank_all_workspace_bytes and blp_rank_all
// cannot disagree.
#pragma once

namespace blp {

enum Knob : int {
    KNOB_RANK_KERNEL = 0,        // 1: every model through the exact f32 kernels whatever the block size
    KNOB_GEMM_KERNEL,            // 1: bilinear pre-pass on the f32 MFMA chain instead of bf16 x 3
    KNOB_SAD_QUERIES_PER_GROUP,  // 16 .. 256 (power of two): queries per workgroup of the TransE pre-pass
    KNOB_SAD_PASS_GROUPS,        // candidate groups per pre-pass slab (forces the multi-slab path)
    KNOB_SAD_MIN_QUERIES,        // smallest block the TransE pre-pass takes
    KNOB_GEMM_PASS_WORDS,        // flag words per query and slab of the bilinear pre-pass (multi-slab path)
    KNOB_GEMM_TILES_PER_CHUNK,   // 16 / 32 / 64 query tiles per workgroup of the bilinear pre-pass
    KNOB_EXACT_QUERY_CHUNK,      // queries per workgroup of the exact f32 kernels (rank_tiles: 16 / 32 / 64 / 128; rank_small: any)
    KNOB_SMALL_KERNEL,           // 1: the small-block exact kernel for every block it can take; 2: never; 3: as 1, TransE on the register-tile kernel
    KNOB_STREAM_KERNEL,          // 2: <= 4 + 4 queries stay on rank_tiles<STATIC> (else: rank_stream.hip); 3 / 4: the workgroup-tile (order-exact keys) / the ring kernel (bilinear models: approximate keys first) whatever the table length; 5 = 3
    KNOB_DKRL_SPLIT,             // 1 / 2 / 4: waves per M-tile of the DKRL table-build kernel (dkrl.hip)
    KNOB_MFMA_SELFTEST,          // 1: the matrix-pipe accumulation self-test (rank_gemm.hip) reports a violation whatever it measured
    KNOB_INBATCH_PROBE,          // timing probe of the in-batch loss forward (wrong results!): bit 0: no index workgroups; bit 1: no tickets / final sum; bit 2: no redundant positives
    KNOB_INBATCH_SHARES,         // waves per entity row of the in-batch loss backward (a power of two <= 16; 0: by batch size)
    KNOB_COUNT
};

#ifdef BLP_TEST_HOOKS
long long knob(int which);  // api.cpp
#else
constexpr long long knob(int) { return 0; }
#endif

}  // namespace blp
