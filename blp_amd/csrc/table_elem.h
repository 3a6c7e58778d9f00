// table_elem.h -- the storage types a CANDIDATE table may come in: f32 (the reference's, train.py:96-97), or the 16-bit copy
// the table build can emit next to it ("keep fp32 (or emit fp16 copy ...)"): IEEE half or bfloat16.  A 16-bit
// element widens to f32 EXACTLY, and every kernel that reads such a table widens first and then runs the f32 arithmetic of the
// reference in the reference's order: the counts are those of the oracle on the widened table, bit for bit.  What the 16-bit
// copy buys is bytes: the few-queries passes over a long table are HBM-bound (rank_stream.hip), and half the bytes is half
// the pass.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace blp {

enum { kTableF32 = 0, kTableF16 = 1, kTableBF16 = 2 };  // == BLP_DTYPE_* (include/blp_hip.h)
__host__ __device__ constexpr int table_elem_bytes(int dtype) { return dtype == kTableF32 ? 4 : 2; }

// the two elements of a 32-bit word of a 16-bit row (element 2j in the low half)
template <class T>
__device__ __forceinline__ void widen_pair(unsigned w, float& lo, float& hi);
template <>
__device__ __forceinline__ void widen_pair<_Float16>(unsigned w, float& lo, float& hi) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    const h2 v = __builtin_bit_cast(h2, w);
    lo = (float)v.x;
    hi = (float)v.y;
}
template <>
__device__ __forceinline__ void widen_pair<__bf16>(unsigned w, float& lo, float& hi) {
    lo = __builtin_bit_cast(float, w << 16);
    hi = __builtin_bit_cast(float, w & 0xffff0000u);
}

// four consecutive elements (16-byte aligned for f32, 8-byte aligned for the 16-bit types) widened
template <class T>
__device__ __forceinline__ float4 load4(const T* p) {
    const uint2 w = *reinterpret_cast<const uint2*>(p);
    float4 v;
    widen_pair<T>(w.x, v.x, v.y);
    widen_pair<T>(w.y, v.z, v.w);
    return v;
}
template <>
__device__ __forceinline__ float4 load4<float>(const float* p) { return *reinterpret_cast<const float4*>(p); }

}  // namespace blp
