// sad_common.h -- pieces shared by the fixed-point TransE pre-pass kernels (rank_sad.hip: D = 64 / 128 / 256
// held in registers; rank_sad_wide.hip: any width, 128 elements at a time): the quantisation map, its
// error accounting and the range reduction's second stage.  See the header of rank_sad.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#pragma clang fp contract(off)

namespace blp {

constexpr int kSW = 4;                       // waves per workgroup
constexpr int kSChunk = 256;                 // queries per workgroup at most (3.75 ms; 128: 3.79, 512: 3.82)
constexpr int kSadMinQueries = 64;          // below this the exact f32 kernels win (tools/bench_small_blocks.py)
constexpr int kSQuota = 1024;                // undecided pairs a workgroup can list
constexpr unsigned kSInvalid = 0x40001000u;  // accumulator bias of padding rows: beyond every T_hi + E_c
constexpr unsigned kSThrMax = 0x3fffffffu;
constexpr unsigned kSNoPair = 0xFFFFFFFFu;

struct SadParams {
    int lo_ord, hi_ord;     // range of table + coefficients, as order-preserving ints
    unsigned maxabs_bits;   // max |value| over table, q_fixed, q_rel (bits of a non-negative float)
    unsigned nonfinite;
    unsigned n_pairs;       // pairs listed in the current pass
    unsigned pad[3];
};
constexpr int kSRangeBlocks = 1024;  // partial results of the range pass

__device__ __forceinline__ int f2ord(float f) { const int b = __float_as_int(f); return b >= 0 ? b : b ^ 0x7fffffff; }
__device__ __forceinline__ float ord2f(int o) { return __int_as_float(o >= 0 ? o : o ^ 0x7fffffff); }

struct SadScale { float lo, scale; bool ok; };

__device__ __forceinline__ SadScale sad_scale(const SadParams* __restrict__ p) {
    SadScale s;
    const int lo_ord = p->lo_ord, hi_ord = p->hi_ord;
    s.lo = ord2f(lo_ord);
    const float range = ord2f(hi_ord) - s.lo;
    s.scale = 65535.0f / range;
    s.ok = !p->nonfinite && lo_ord <= hi_ord && range > 0.f && range < 3.0e38f && s.scale > 0.f && s.scale < 3.0e38f;
    return s;
}

// x^ = rint(v), v = fl(fl(x - lo) s): |v - (x - lo) s| <= 65535 (2u + u^2) < 0.0079 for every x in
// [lo, hi], and v - x^ is exact in f32, so |x^ - (x - lo) s| <= resid + 0.0079 with resid += |v - x^|.
__device__ __forceinline__ unsigned sad_quant(float x, const SadScale& s, float& resid) {
    const float v = fminf(fmaxf((x - s.lo) * s.scale, 0.f), 65535.f);
    const float r = rintf(v);
    resid += fabsf(v - r);
    return (unsigned)r;
}
constexpr float kSResidSlack = 0.0079f;  // per element, see sad_quant

__device__ __forceinline__ float sad_coef(float fixed, float rel, bool head) { return head ? fixed - rel : fixed + rel; }

static __global__ __launch_bounds__(64) void sad_range_finish_kernel(const SadParams* __restrict__ partial, int n,
                                                              SadParams* __restrict__ out) {
    int lo = 0x7fffffff, hi = (int)0x80000000;
    unsigned m = 0, bad = 0;
    for (int i = threadIdx.x; i < n; i += 64) {
        const SadParams p = partial[i];
        lo = p.lo_ord < lo ? p.lo_ord : lo;
        hi = p.hi_ord > hi ? p.hi_ord : hi;
        m = p.maxabs_bits > m ? p.maxabs_bits : m;  // non-negative floats order like their bits
        bad |= p.nonfinite;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const int l2 = __shfl_xor(lo, off), h2 = __shfl_xor(hi, off);
        const unsigned m2 = __shfl_xor(m, off);
        lo = l2 < lo ? l2 : lo;
        hi = h2 > hi ? h2 : hi;
        m = m2 > m ? m2 : m;
        bad |= __shfl_xor(bad, off);
    }
    if (threadIdx.x == 0) {
        SadParams p = {};
        p.lo_ord = lo; p.hi_ord = hi; p.maxabs_bits = m; p.nonfinite = bad;
        *out = p;
    }
}

}  // namespace blp
