// bow.hip -- the entity-table build for the bag-of-words encoder (models.py:143-155, the `glove-bow` / `bert-bow` script
// families), fused with the steps that follow it:
//     embs = self.embeddings(text_tok)                                     models.py:150   (B, L, E) gathered
//     lengths = torch.sum(text_mask, dim=-1, keepdim=True)                 models.py:151
//     embs = torch.sum(text_mask.unsqueeze(dim=-1) * embs, dim=1)          models.py:152   a second (B, L, E) temporary
//     embs = embs / lengths                                                models.py:153
//     ent_emb = F.normalize(ent_emb, dim=-1)                               models.py:40-41 (TransE: always, for these scripts)
//     ent_emb[idx:idx + batch] = batch_emb                                 train.py:109-113
// out[i, :] = (sum_l mask[i, l] * emb[tok[i, l], :]) / (sum_l mask[i, l])  [ / max(||.||_2, 1e-12) ], written straight into
// the caller's rows of the table shard.  The stock modules stream the (B, L, E) gather three times (write, read + write of
// the masked product, read); here every gathered embedding row is read once and nothing is materialised: the kernel is
// bound by that gather (E x 4 bytes per token; the BERT word-embedding table, 28 996 x 768 f32 = 89 MB, sits in the
// Infinity Cache, the 480 MB GloVe table does not).
// One WAVE per entity: lane l owns columns 4 l .. 4 l + 3 of every 256-column sweep (E <= 1024: four sweeps), the tokens
// are walked in order, FOUR tokens' rows requested together; product and sum are rounded separately, tokens in order (the
// reference's own order is whatever its reduction kernel does on the device it runs on: floating point with a tolerance).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "launch.h"

#pragma clang fp contract(off)

namespace blp {

constexpr int kBowMaxE = 1024;
constexpr int kBowSweeps = kBowMaxE / 256;

__global__ __launch_bounds__(256) void bow_rows_kernel(const int64_t* __restrict__ tok, const float* __restrict__ mask, int64_t n,
                                                       int L, const float* __restrict__ emb, int64_t V, int E, int normalize,
                                                       float* __restrict__ out, int64_t ldo, int* __restrict__ bad_tok) {
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;  // (wave-uniform; nothing below synchronises the workgroup)
    const int64_t* t_row = tok + i * L;
    const float* m_row = mask ? mask + i * L : nullptr;
    float4 acc[kBowSweeps];
#pragma unroll
    for (int k = 0; k < kBowSweeps; ++k) acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    float length = 0.f;
    bool bad = false;
    for (int l0 = 0; l0 < L; l0 += 4) {
        float4 v[4][kBowSweeps];
        float m[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {  // four tokens' rows in flight
            const int l = l0 + j;
            int64_t t = l < L ? t_row[l] : 0;
            m[j] = l < L ? (m_row ? m_row[l] : 1.0f) : 0.0f;
            if ((uint64_t)t >= (uint64_t)V) { bad |= l < L; t = 0; }  // (nn.Embedding would raise: flagged, row 0 read instead)
            const float* row = emb + t * (int64_t)E;
#pragma unroll
            for (int k = 0; k < kBowSweeps; ++k) {
                const int c = 4 * lane + 256 * k;
                v[j][k] = (l < L && c < E) ? *reinterpret_cast<const float4*>(row + c) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (l0 + j < L) {  // wave-uniform
                length = length + m[j];
#pragma unroll
                for (int k = 0; k < kBowSweeps; ++k) {
                    const float px = m[j] * v[j][k].x, py = m[j] * v[j][k].y, pz = m[j] * v[j][k].z, pw = m[j] * v[j][k].w;
                    acc[k].x = acc[k].x + px; acc[k].y = acc[k].y + py; acc[k].z = acc[k].z + pz; acc[k].w = acc[k].w + pw;
                }
            }
        }
    }
    float ss = 0.f;
#pragma unroll
    for (int k = 0; k < kBowSweeps; ++k) {
        acc[k].x = acc[k].x / length; acc[k].y = acc[k].y / length; acc[k].z = acc[k].z / length; acc[k].w = acc[k].w / length;
        if (4 * lane + 256 * k < E) ss += acc[k].x * acc[k].x + acc[k].y * acc[k].y + acc[k].z * acc[k].z + acc[k].w * acc[k].w;
    }
    float scale = 1.0f;
    if (normalize) {  // F.normalize: x / max(||x||_2, 1e-12)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off);
        const float nrm = sqrtf(ss);
        scale = nrm > 1e-12f ? nrm : 1e-12f;  // (divided by, as F.normalize does: a multiplication by the reciprocal differs in the last bit)
    }
    float* dst = out + i * ldo;
#pragma unroll
    for (int k = 0; k < kBowSweeps; ++k) {
        const int c = 4 * lane + 256 * k;
        if (c < E) {
            if (normalize)
                *reinterpret_cast<float4*>(dst + c) = make_float4(acc[k].x / scale, acc[k].y / scale, acc[k].z / scale, acc[k].w / scale);
            else
                *reinterpret_cast<float4*>(dst + c) = acc[k];
        }
    }
    if (bad && lane == 0) atomicMin(bad_tok, -1);
}

bool bow_rows_supported(int E) { return E > 0 && E % 4 == 0 && E <= kBowMaxE; }

hipError_t launch_bow_rows(const int64_t* tok, const float* mask, int64_t n, int L, const float* emb, int64_t V, int E,
                           int normalize, float* out, int64_t ldo, int* bad_tok, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    const int64_t blocks = (n + 3) / 4;
    if (blocks > 0x7fffffff) return hipErrorInvalidValue;
    bow_rows_kernel<<<dim3((unsigned)blocks), 256, 0, stream>>>(tok, mask, n, L, emb, V, E, normalize, out, ldo, bad_tok);
    return hipGetLastError();
}

}  // namespace blp
