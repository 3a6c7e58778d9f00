// dot_band.h -- what the bilinear models' approximate-score paths share (rank_gemm.hip: the MFMA pre-passes; rank_stream.hip:
// a few queries against a long table): the query-side operand W_q of score(q, c) = <W_q, e_c>, its absolute bound B_q, and
// the band factors that turn the two into eps(q, c) >= |S~ - S_ref| (rank_gemm.hip's header has the derivation).
#pragma once
#include <hip/hip_runtime.h>

#include "score_core.h"

#pragma clang fp contract(off)

namespace blp {

// f32 evaluation of <W_q, e_c> as one chain of <= 128 fused multiply-adds (or exact-product MFMA steps): the chain errs by
// <= n u T, the reference by <= (n + 2) u T, the operands by 3 u T, T = sum_k B_q[k] |e_c[k]| <= ||B_q|| ||e_c||:
// 2 (n + 2) + slack for n <= 128 terms.
constexpr float kBandC = 320.0f;

// GEMM operand w and its absolute bound b for element k of a query (f = the fixed entity, r = relation).
template <int MODEL, int SIDE>
__device__ __forceinline__ void gemm_operand(const float* __restrict__ f, const float* __restrict__ r, int k, int D,
                                             float& w, float& b) {
    const int H = D / 2, j = k < H ? k : k - H;
    if constexpr (MODEL == DISTMULT) {
        w = f[k] * r[k];
        b = fabsf(w);
    } else if constexpr (MODEL == COMPLEX) {
        const float fr = f[j], fi = f[H + j], rr = r[j], ri = r[H + j];
        float p, q;
        if (SIDE == TAIL) {  // f = head:  re: rr*hr - ri*hi ; im: rr*hi + ri*hr
            if (k < H) { p = rr * fr; q = ri * fi; w = p - q; } else { p = rr * fi; q = ri * fr; w = p + q; }
        } else {             // f = tail:  re: rr*tr + ri*ti ; im: rr*ti - ri*tr
            if (k < H) { p = rr * fr; q = ri * fi; w = p + q; } else { p = rr * fi; q = ri * fr; w = p - q; }
        }
        b = fabsf(p) + fabsf(q);
    } else {  // SIMPLE: f = [f_head_role | f_tail_role], r = [ra | rb]
        if (SIDE == TAIL) w = k < H ? r[H + j] * f[H + j] : f[j] * r[j];      // cand [th | tt]: rb*ht , hh*ra
        else              w = k < H ? r[j] * f[H + j] : f[j] * r[H + j];      // cand [hh | ht]: ra*tt , th*rb
        w = w * 0.5f;
        b = fabsf(w);
    }
}

// band factor of a row from its norm / largest magnitude (see above); exact zero rows need no band
__device__ __forceinline__ float band_norm(float sumsq, float maxabs) {
    if (maxabs == 0.f) return 0.f;
    if (!(maxabs >= 1e-18f && maxabs <= 3.0e38f)) return __builtin_inff();  // also NaN
    return sqrtf(sumsq) * 1.0001f;  // overflow of sumsq gives inf: conservative
}

}  // namespace blp
