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
constexpr int kSadMinQueries = 64;          // below this, or below kSadMinPairs (query, candidate) pairs, the exact
constexpr long long kSadMinPairs = 4000000;  // f32 kernels win (tools/bench_small_blocks.py)
constexpr int kSQuota = 1024;                // undecided pairs a workgroup can list
constexpr unsigned kSInvalid = 0x40001000u;  // accumulator bias of padding rows: beyond every T_hi + E_c
constexpr unsigned kSThrMax = 0x3fffffffu;
constexpr unsigned kSNoPair = 0xFFFFFFFFu;

constexpr unsigned kSRowExact = 0xFFFFFFFFu;  // resid[] of a row the pre-pass must not decide (see below)

// Range pass result.  The quantisation range [lo, hi] is that of the FINITE table values and query coefficients,
// clamped to mean +- 16 standard deviations so that an isolated outlier cannot stretch it (and with it the band of
// every pair).  What falls outside -- non-finite values, outliers -- is not quantised at all: a table row holding
// such a value is marked "exact only" (resid[row] = kSRowExact: undecided against every query), a query whose
// coefficients hold one gets thresholds that decide nothing.  The result stays exact whatever the data; only the
// marked rows / queries take the slow path (one Inf in the FB15k-237 table: 121 ms with the old global off-switch).
struct SadParams {
    int lo_ord, hi_ord;     // range, as order-preserving ints
    unsigned nonfinite;     // number of non-finite values seen (information only)
    unsigned n_pairs;       // pairs listed in the current pass
    double sum, sumsq;      // of the finite values (partial results; the final record holds the totals)
    unsigned long long count;
};
constexpr int kSRangeBlocks = 1024;  // partial results of the range pass
constexpr float kSClampSigmas = 16.0f;

__device__ __forceinline__ int f2ord(float f) { const int b = __float_as_int(f); return b >= 0 ? b : b ^ 0x7fffffff; }
__device__ __forceinline__ float ord2f(int o) { return __int_as_float(o >= 0 ? o : o ^ 0x7fffffff); }

struct SadScale { float lo, hi, scale; bool ok; };

__device__ __forceinline__ SadScale sad_scale(const SadParams* __restrict__ p) {
    SadScale s;
    const int lo_ord = p->lo_ord, hi_ord = p->hi_ord;
    s.lo = ord2f(lo_ord);
    s.hi = ord2f(hi_ord);
    const float range = s.hi - s.lo;
    s.scale = 65535.0f / range;
    s.ok = lo_ord <= hi_ord && range > 0.f && range < 3.0e38f && s.scale > 0.f && s.scale < 3.0e38f;
    return s;
}
// largest |value| a quantised (in-range) element can have: bounds the table side of the rounding term rho
__device__ __forceinline__ float sad_range_maxabs(const SadScale& s) { return fmaxf(fabsf(s.lo), fabsf(s.hi)); }

// x^ = rint(v), v = fl(fl(x - lo) s): |v - (x - lo) s| <= 65535 (2u + u^2) < 0.0079 for every x in
// [lo, hi], and v - x^ is exact in f32, so |x^ - (x - lo) s| <= resid + 0.0079 with resid += |v - x^|.
// `outside` collects values the map does not cover (NaN, Inf, beyond the clamped range): their row / query is
// marked exact-only by the caller.
__device__ __forceinline__ unsigned sad_quant(float x, const SadScale& s, float& resid, bool& outside) {
    outside |= !(x >= s.lo && x <= s.hi);  // also NaN
    const float v = fminf(fmaxf((x - s.lo) * s.scale, 0.f), 65535.f);
    const float r = rintf(v);
    resid += fabsf(v - r);
    return (unsigned)r;
}
constexpr float kSResidSlack = 0.0079f;  // per element, see sad_quant

__device__ __forceinline__ float sad_coef(float fixed, float rel, bool head) { return head ? fixed - rel : fixed + rel; }

// One thread's view of the range pass; merge() combines two, reduce_store() reduces a 256-thread block.
struct SadRange {
    int lo = 0x7fffffff, hi = (int)0x80000000;
    unsigned bad = 0;
    double sum = 0.0, sumsq = 0.0;
    unsigned long long count = 0;
    __device__ __forceinline__ void see(float x) {
        if (!(fabsf(x) < 3.0e38f)) { ++bad; return; }  // NaN, Inf or too close to overflow
        const int o = f2ord(x);
        lo = o < lo ? o : lo;
        hi = o > hi ? o : hi;
        sum += (double)x;
        sumsq += (double)x * (double)x;
        ++count;
    }
    __device__ __forceinline__ void merge(int lo2, int hi2, unsigned bad2, double sum2, double sumsq2, unsigned long long count2) {
        lo = lo2 < lo ? lo2 : lo;
        hi = hi2 > hi ? hi2 : hi;
        bad += bad2; sum += sum2; sumsq += sumsq2; count += count2;
    }
    __device__ __forceinline__ void wave_reduce() {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1)
            merge(__shfl_xor(lo, off), __shfl_xor(hi, off), __shfl_xor(bad, off), __shfl_xor(sum, off), __shfl_xor(sumsq, off),
                  __shfl_xor(count, off));
    }
    __device__ __forceinline__ SadParams record() const {
        SadParams p = {};
        p.lo_ord = lo; p.hi_ord = hi; p.nonfinite = bad; p.sum = sum; p.sumsq = sumsq; p.count = count;
        return p;
    }
};

// blockDim.x == 256: partial[blockIdx.x] = the block's result (fixed order: reproducible)
__device__ __forceinline__ void sad_range_block_store(SadRange r, SadParams* __restrict__ partial) {
    __shared__ SadParams part[4];
    r.wave_reduce();
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = r.record();
    __syncthreads();
    if (threadIdx.x == 0) {
        SadRange t;
        for (int w = 0; w < 4; ++w) t.merge(part[w].lo_ord, part[w].hi_ord, part[w].nonfinite, part[w].sum, part[w].sumsq, part[w].count);
        partial[blockIdx.x] = t.record();
    }
}

// The range of the whole input from the n partial results, by one wave (fixed order: every caller gets the same bits);
// the result is valid on lane 0.
__device__ __forceinline__ SadParams sad_range_finish_wave(const SadParams* __restrict__ partial, int n, int lane) {
    SadRange r;
    for (int i = lane; i < n; i += 64) {
        const SadParams p = partial[i];
        r.merge(p.lo_ord, p.hi_ord, p.nonfinite, p.sum, p.sumsq, p.count);
    }
    r.wave_reduce();
    SadParams p = r.record();
    p.n_pairs = 0;
    if (r.count > 0 && r.lo <= r.hi) {  // clamp the range to mean +- kSClampSigmas standard deviations
        const double mean = r.sum / (double)r.count;
        const double var = r.sumsq / (double)r.count - mean * mean;
        const double sd = var > 0.0 ? sqrt(var) : 0.0;
        const float lo_c = (float)(mean - kSClampSigmas * sd), hi_c = (float)(mean + kSClampSigmas * sd);
        if (lo_c > ord2f(r.lo)) p.lo_ord = f2ord(lo_c);
        if (hi_c < ord2f(r.hi)) p.hi_ord = f2ord(hi_c);
    }
    return p;
}

static __global__ __launch_bounds__(64) void sad_range_finish_kernel(const SadParams* __restrict__ partial, int n,
                                                              SadParams* __restrict__ out) {
    const SadParams p = sad_range_finish_wave(partial, n, threadIdx.x);
    if (threadIdx.x == 0) *out = p;
}

}  // namespace blp
