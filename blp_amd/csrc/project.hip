// project.hip -- the last step of the entity-table build for the BERT encoders, fused:
//     embs = self.enc_linear(embs)              models.py:110-111  (nn.Linear(hidden_size, dim, bias=False), :104)
//     ent_emb = F.normalize(ent_emb, dim=-1)    models.py:40-41    (TransE only)
//     ent_emb[idx:idx + batch] = batch_emb      train.py:109-113
// out[i, :] = (x[i, :] . W^T) [/ max(||.||_2, 1e-12)], written straight into the caller's rows of the table shard.
//
// GEMM-shaped, so it runs on the matrix cores -- v_mfma_f32_32x32x2_f32: f32 operands, f32 accumulation, no reduced
// precision anywhere (the table is the input of the exact ranking kernels).  A workgroup of EIGHT waves owns 64 rows x
// all D output columns: waves 0-3 the upper 32 rows, 4-7 the lower, each the 32-column blocks w % 4, w % 4 + 4, ...
// The K = hidden_size loop runs 32 elements per step through a double-buffered LDS stage (row stride 33 floats:
// conflict-free per-lane operand reads) fed from a two-deep register ring: the global loads of step k + 3 are issued
// before the MFMAs of step k and written to LDS after step k + 1, one barrier per step.  The accumulators go through
// LDS once more for the row norms and leave as whole rows.
// Floating point with a tolerance (tests: 2e-6 relative to the f64 result): the reduction order over K differs from
// any BLAS's, as BLAS libraries differ from each other; the table is then the ranking's exact input either way.
// [measured] 14 541 x 768 -> 128 with normalisation: see DESIGN.md 4.8.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "launch.h"

#pragma clang fp contract(off)

namespace blp {

constexpr int kPM = 64;        // rows per workgroup
constexpr int kPK = 32;        // K elements staged per step
constexpr int kPS = 33;        // LDS row stride (floats)
constexpr int kPThreads = 512;
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int D>
__global__ __launch_bounds__(kPThreads) void project_rows_kernel(const float* __restrict__ x, int64_t n, int64_t ldx,
                                                                 const float* __restrict__ w, int E, int normalize,
                                                                 float* __restrict__ out, int64_t ldo) {
    constexpr int NB = D / 32;                 // 32-column blocks
    constexpr int NBW = (NB + 3) / 4;          // blocks per wave
    constexpr int WP = D * 8 / kPThreads;      // 16-byte pieces of the W stage per thread (the x stage: one)
    constexpr int STAGE = (kPM + D) * kPS;     // floats per LDS buffer
    static_assert(kPM * 8 == kPThreads && D * 8 % kPThreads == 0, "staging plan");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int mb = wave >> 2, wq = wave & 3;
    const int64_t row0 = (int64_t)blockIdx.x * kPM;
    const int n_steps = (E + kPK - 1) / kPK;

    // this thread's pieces: row pr, floats pc .. pc + 3 of a 32-float stage row
    const int pr = tid >> 3, pc = (tid & 7) * 4;
    int64_t xrow = row0 + pr;
    xrow = xrow < n ? xrow : n - 1;  // rows past the end: clamped, never stored
    const float* xsrc = x + xrow * ldx + pc;
    const float* wsrc = w + (int64_t)pr * E + pc;

    float4 rx[2], rw[2][WP];
    auto fetch = [&](int step, float4& fx, float4 (&fw)[WP]) {  // K tail (E % 32): zeros; E % 4 == 0
        const int k0 = step * kPK;
        const bool live = step < n_steps && k0 + pc < E;
        fx = live ? *reinterpret_cast<const float4*>(xsrc + k0) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int i = 0; i < WP; ++i)
            fw[i] = live ? *reinterpret_cast<const float4*>(wsrc + (int64_t)i * (kPThreads / 8) * E + k0) : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    auto put = [&](int buf, const float4& fx, const float4 (&fw)[WP]) {
        float* xs = smem + buf * STAGE;
        float* ws = xs + kPM * kPS;
        float* d = xs + pr * kPS + pc;
        d[0] = fx.x; d[1] = fx.y; d[2] = fx.z; d[3] = fx.w;
#pragma unroll
        for (int i = 0; i < WP; ++i) {
            float* e = ws + (pr + i * (kPThreads / 8)) * kPS + pc;
            e[0] = fw[i].x; e[1] = fw[i].y; e[2] = fw[i].z; e[3] = fw[i].w;
        }
    };

    f32x16 acc[NBW];
#pragma unroll
    for (int b = 0; b < NBW; ++b)
        acc[b] = (f32x16){0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    auto compute = [&](int buf) {  // 32x32x2: lane l supplies A[l % 32][l / 32] and B[l / 32][l % 32] of the slice
        const float* xs = smem + buf * STAGE;
        const float* ws = xs + kPM * kPS;
#pragma unroll
        for (int kp = 0; kp < kPK / 2; ++kp) {
            const int k = 2 * kp + (lane >> 5);
            const float a = xs[(32 * mb + (lane & 31)) * kPS + k];
#pragma unroll
            for (int b = 0; b < NBW; ++b) {
                const int nb = wq + 4 * b;
                if (nb < NB) acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, ws[(32 * nb + (lane & 31)) * kPS + k], acc[b], 0, 0, 0);
            }
        }
    };

    // prologue: steps 0 and 1 in LDS, steps 2 and 3 on their way in the register ring
    fetch(0, rx[0], rw[0]);
    fetch(1, rx[1], rw[1]);
    put(0, rx[0], rw[0]);
    put(1, rx[1], rw[1]);
    fetch(2, rx[0], rw[0]);
    fetch(3, rx[1], rw[1]);
    __syncthreads();
    // step k: MFMAs from LDS[k & 1]; barrier; ring slot k & 1 (step k + 2) -> LDS[k & 1]; slot refilled with step k + 4
    for (int k = 0; k < n_steps; k += 2) {
        compute(0);
        __syncthreads();
        put(0, rx[0], rw[0]);
        fetch(k + 4, rx[0], rw[0]);
        if (k + 1 < n_steps) compute(1);
        __syncthreads();
        put(1, rx[1], rw[1]);
        fetch(k + 5, rx[1], rw[1]);
    }
    __syncthreads();

    // accumulator register j of lane l: row (j % 4) + 8 (j / 4) + 4 (l / 32) of the 32-row block, column l % 32
    constexpr int OS = D + 1;
    float* os = smem;  // kPM x OS, over the stages
#pragma unroll
    for (int b = 0; b < NBW; ++b) {
        const int nb = wq + 4 * b;
        if (nb < NB) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int r = 32 * mb + (j & 3) + 8 * (j >> 2) + 4 * (lane >> 5);
                os[r * OS + 32 * nb + (lane & 31)] = acc[b][j];
            }
        }
    }
    __syncthreads();
    // eight threads per row: sum of squares of an eighth each, combined inside the octet
    const int r = tid >> 3, part = tid & 7;
    float scale = 1.0f;
    if (normalize) {
        float ss = 0.0f;
        for (int c = part; c < D; c += 8) ss += os[r * OS + c] * os[r * OS + c];
        ss += __shfl_xor(ss, 1);
        ss += __shfl_xor(ss, 2);
        ss += __shfl_xor(ss, 4);
        const float nrm = sqrtf(ss);
        scale = 1.0f / (nrm > 1e-12f ? nrm : 1e-12f);  // F.normalize: x / max(||x||, eps), eps = 1e-12
    }
    if (row0 + r < n) {
        float* dst = out + (row0 + r) * ldo;
        for (int c = part; c < D; c += 8) dst[c] = normalize ? os[r * OS + c] * scale : os[r * OS + c];
    }
}

bool project_rows_supported(int E, int D) { return E > 0 && E % 4 == 0 && (D == 64 || D == 128 || D == 256); }

hipError_t launch_project_rows(const float* x, int64_t n, int64_t ldx, const float* w, int E, int D, int normalize,
                               float* out, int64_t ldo, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    const unsigned blocks = (unsigned)((n + kPM - 1) / kPM);
    const size_t stage = (size_t)2 * (kPM + D) * kPS * 4, epi = (size_t)kPM * (D + 1) * 4;
    const size_t lds = stage > epi ? stage : epi;
#define BLP_PROJECT_CASE(DD)                                                                                          \
    case DD: {                                                                                                        \
        if (lds > 64 * 1024) {  /* beyond the default dynamic-LDS limit */                                           \
            const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&project_rows_kernel<DD>),          \
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);            \
            if (e != hipSuccess) return e;                                                                            \
        }                                                                                                             \
        project_rows_kernel<DD><<<blocks, kPThreads, lds, stream>>>(x, n, ldx, w, E, normalize, out, ldo);            \
        break;                                                                                                        \
    }
    switch (D) {
        BLP_PROJECT_CASE(64)
        BLP_PROJECT_CASE(128)
        BLP_PROJECT_CASE(256)
    default: return hipErrorInvalidValue;
    }
#undef BLP_PROJECT_CASE
    return hipGetLastError();
}

}  // namespace blp
