// dkrl.hip -- the entity-table build for the DKRL encoder (models.py:158-204, the `bert-dkrl` / `glove-dkrl` script
// families), fused with the steps that follow it:
//     embs = self.embeddings(text_tok) * text_mask.unsqueeze(-1)               models.py:177   (B, L, E) gathered
//     embs = self.conv1(F.pad(embs.transpose(1, 2), [0, 1])) * text_mask        models.py:180-187  Conv1d(E, dim, 2)
//     embs = F.max_pool1d(embs, 4); text_mask = F.max_pool1d(text_mask, 4)      models.py:188-195  (L >= 4: kernel size 4)
//     embs = self.conv2(F.pad(torch.tanh(embs), [0, 1]))                        models.py:196-198  Conv1d(dim, dim, 2)
//     embs = torch.tanh(torch.sum(embs * text_mask, dim=-1) / lengths)          models.py:199-202
//     ent_emb = F.normalize(ent_emb, dim=-1)                                    models.py:40-41 (TransE: always, for these scripts)
//     ent_emb[idx:idx + batch] = batch_emb                                      train.py:109-113
// The stock modules run it as a gather, two transposes, two MIOpen convolutions, two poolings and five elementwise
// kernels over (B, L, E) / (B, dim, L) temporaries: [measured] 8.5 ms for the FB15k-237 table at E = 768 (22 TF/s of conv
// arithmetic).  Here an entity never leaves the chip between its token ids and its table row.
//
// conv1 is a GEMM -- H1[l, c] = b1[c] + sum_e X[l, e] W1[c, e, 0] + X[l + 1, e] W1[c, e, 1], X[l] = mask[l] emb[tok[l]],
// X[L] = 0 -- and runs on the matrix cores with f32 operands (v_mfma_f32_32x32x2_f32: no reduced precision; the table is the
// input of the exact ranking kernels): a wave owns ONE M-tile of 32 token positions x all 128 output channels (four
// accumulator blocks), the second tap is the same A rows shifted by one.  A workgroup of four waves = four M-tiles (four
// entities of up to 32 tokens, or two of up to 64) that walk E together, 32 columns per step, through a double-buffered LDS
// stage: the W1 slice (128 x 32 x 2 taps, shared by the four waves; row stride 66 floats: both taps of a (channel, e) in one
// conflict-free ds_read_b64) and each wave's own 33 gathered rows (row stride 33), fed from a register ring two steps deep
// as in project.hip.  In the accumulator layout the four positions of a max-pool window are four registers of one lane, so
// bias, mask, max-pool and tanh happen in registers.  conv2 and the masked mean are linear in the pooled activations T:
//     sum_j pm[j] (b2[c] + sum_d W2[c, d, 0] T[j, d] + W2[c, d, 1] T[j + 1, d]) = b2[c] S + sum_d W2[c, d, 0] U0[d] + W2[c, d, 1] U1[d]
// with U0[d] = sum_j pm[j] T[j, d], U1[d] = sum_j pm[j] T[j + 1, d], S = sum_j pm[j] (pm = the pooled mask): two 128-vectors
// per entity through LDS, then 256 multiply-adds per output channel -- conv2 never exists as a (J, 128) array.
// Floating point with a tolerance (tests: against the stock modules in float64): the reduction orders differ from
// MIOpen's, as convolution libraries differ from each other; the table is then the ranking's exact input either way.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "knobs.h"
#include "launch.h"

#pragma clang fp contract(off)

namespace blp {

constexpr int kKD = 128;       // output channels (dim: every DKRL script uses the default 128)
constexpr int kKStep = 32;     // embedding columns per stage
constexpr int kKXS = 33;       // LDS row stride of a wave's gathered rows (floats)
constexpr int kKWS = 66;       // LDS row stride of the W1 slice (floats): 32 e x 2 taps + 2
constexpr int kKMaxL = 64;
constexpr int kKXFloats = 33 * kKXS, kKWFloats = kKD * kKWS;
constexpr int kKStage = 4 * kKXFloats + kKWFloats;  // floats per LDS buffer
typedef float kf32x16 __attribute__((ext_vector_type(16)));

// NSPLIT waves share an M-tile, each with 4 / NSPLIT of its channel blocks: 4 / NSPLIT M-tiles per workgroup.  A table chunk
// of emb_batch_size = 512 entities (scripts/*-fb15k237.sh) is 128 workgroups at NSPLIT = 1 -- half the chip; the launcher picks
// the split that fills it (the W1 slice is then shared by fewer entities: more L2 traffic, the same arithmetic).
template <int NSPLIT>
__global__ __launch_bounds__(256) void dkrl_rows_kernel(const int64_t* __restrict__ tok, const float* __restrict__ mask, int64_t n,
                                                        int L, const float* __restrict__ emb, int64_t V, int E,
                                                        const float* __restrict__ w1, const float* __restrict__ b1,
                                                        const float* __restrict__ w2, const float* __restrict__ b2, int normalize,
                                                        float* __restrict__ out, int64_t ldo, int* __restrict__ bad_tok) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ float smask[4][kKMaxL + 4];  // per local entity: its mask row, zero-padded
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int MT = 4 / NSPLIT, NB = 4 / NSPLIT;              // M-tiles per workgroup; channel blocks per wave
    const int tiles = (L + 31) / 32, epw = MT / tiles;           // M-tiles per entity (1 or 2), entities per workgroup
    const int mt = wave / NSPLIT, nb0 = (wave % NSPLIT) * NB;    // this wave's M-tile of the workgroup, its first channel block
    const bool gathers = wave % NSPLIT == 0;                     // the wave that fetches the M-tile's rows
    const int ent = mt / tiles, tile = mt % tiles;               // the M-tile's entity (local) and its place in it
    const int64_t i_real = (int64_t)blockIdx.x * epw + ent, i = i_real < n ? i_real : n - 1;
    const int n_steps = (E + kKStep - 1) / kKStep;

    for (int x = tid; x < 4 * (kKMaxL + 4); x += 256) {
        const int e = x / (kKMaxL + 4), l = x % (kKMaxL + 4);
        const int64_t ie = (int64_t)blockIdx.x * epw + e;
        smask[e][l] = (e < epw && l < L) ? (mask ? mask[(ie < n ? ie : n - 1) * L + l] : 1.0f) : 0.0f;
    }

    // this lane's pieces of the wave's 33 gathered rows: rows (lane >> 3) + 8 j (j < 4) and, lanes 0 .. 7, row 32; 16 B each
    const int pc = (lane & 7) * 4;
    const float* xrow[5];
    float xm[5];
    bool bad = false;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const int r = j < 4 ? (lane >> 3) + 8 * j : 32, l = 32 * tile + r;
        const bool live = l < L && (j < 4 || lane < 8);
        int64_t t = live ? tok[i * L + l] : 0;
        if ((uint64_t)t >= (uint64_t)V) { bad |= live; t = 0; }  // (nn.Embedding would raise: flagged, row 0 read instead)
        xrow[j] = emb + t * (int64_t)E + pc;
        xm[j] = live ? (mask ? mask[i * L + l] : 1.0f) : 0.0f;   // rows past the description: zeros (F.pad, and the M-tile's tail)
    }
    // ... and of the W1 slice: channels (tid >> 4) + 16 j (j < 8), floats (tid & 15) * 4 .. + 3 of the slice's 64 (e-major, taps inside)
    const int wc = (tid & 15) * 4;
    const float* wsrc = w1 + (int64_t)(tid >> 4) * E * 2 + wc;

    float4 rx[2][5], rw[2][8];
    auto fetch = [&](int step, float4 (&fx)[5], float4 (&fw)[8]) {  // E tail (E % 32): zeros; E % 4 == 0
        const int e0 = step * kKStep;
        const bool xlive = step < n_steps && e0 + pc < E, wlive = step < n_steps && e0 + wc / 2 < E;
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gathers && xlive && xm[j] != 0.0f) v = *reinterpret_cast<const float4*>(xrow[j] + e0);
            fx[j] = make_float4(v.x * xm[j], v.y * xm[j], v.z * xm[j], v.w * xm[j]);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j)
            fw[j] = wlive ? *reinterpret_cast<const float4*>(wsrc + (int64_t)j * 16 * E * 2 + e0 * 2) : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    auto put = [&](int buf, const float4 (&fx)[5], const float4 (&fw)[8]) {
        float* xs = smem + buf * kKStage + mt * kKXFloats;
        float* ws = smem + buf * kKStage + 4 * kKXFloats;
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            if (gathers && (j < 4 || lane < 8)) {
                float* d = xs + (j < 4 ? (lane >> 3) + 8 * j : 32) * kKXS + pc;
                d[0] = fx[j].x; d[1] = fx[j].y; d[2] = fx[j].z; d[3] = fx[j].w;
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float* d = ws + ((tid >> 4) + 16 * j) * kKWS + wc;
            *reinterpret_cast<float2*>(d) = make_float2(fw[j].x, fw[j].y);
            *reinterpret_cast<float2*>(d + 2) = make_float2(fw[j].z, fw[j].w);
        }
    };

    kf32x16 acc[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b)
        acc[b] = (kf32x16){0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    auto compute = [&](int buf) {  // 32x32x2: lane l supplies A[l % 32][l / 32] and B[l / 32][l % 32] of the k-slice
        const float* xs = smem + buf * kKStage + mt * kKXFloats + (lane & 31) * kKXS + (lane >> 5);
        const float* ws = smem + buf * kKStage + 4 * kKXFloats + (32 * nb0 + (lane & 31)) * kKWS + 2 * (lane >> 5);
#pragma unroll
        for (int kp = 0; kp < kKStep / 2; ++kp) {
            const float a0 = xs[2 * kp], a1 = xs[kKXS + 2 * kp];  // X[l][e], X[l + 1][e]
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const float2 w = *reinterpret_cast<const float2*>(ws + 32 * b * kKWS + 4 * kp);  // W1[c][e][0], W1[c][e][1]
                acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, w.x, acc[b], 0, 0, 0);
                acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, w.y, acc[b], 0, 0, 0);
            }
        }
    };

    fetch(0, rx[0], rw[0]);
    fetch(1, rx[1], rw[1]);
    put(0, rx[0], rw[0]);
    put(1, rx[1], rw[1]);
    fetch(2, rx[0], rw[0]);
    fetch(3, rx[1], rw[1]);
    __syncthreads();
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

    // accumulator register j of lane l: position 32 tile + (j % 4) + 8 (j / 4) + 4 (l / 32), channel 32 b + l % 32.
    // bias, mask, max-pool over the four positions of a window (= the four registers 4 g .. 4 g + 3), tanh; pooled position
    // 8 tile + 2 g + l / 32.  T[ent][jp][c] over the stages (row stride 129).
    constexpr int TS = kKD + 1, kJMax = kKMaxL / 4;
    float* ts = smem;                       // 4 x kJMax x TS
    float* us = smem + 4 * kJMax * TS;      // 4 x 2 x kKD
    float* spm = us + 4 * 2 * kKD;          // 4 x kJMax pooled masks, then 4 sums
    const int J = L / 4;                    // F.max_pool1d floors
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int c = 32 * (nb0 + b) + (lane & 31);
        const float bias = b1[c];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int jp = 8 * tile + 2 * g + (lane >> 5), l0 = 4 * jp;
            float p = -__builtin_inff();
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float h = (acc[b][4 * g + q] + bias) * smask[ent][l0 + q];
                p = h > p ? h : p;  // (a NaN never wins here; F.max_pool1d propagates it: the tolerance tests use finite data)
            }
            if (jp < J) ts[(ent * kJMax + jp) * TS + c] = tanhf(p);
        }
    }
    if (tid < 4 * kJMax) {
        const int e = tid / kJMax, jp = tid % kJMax;
        float pm = 0.0f;
        for (int q = 0; q < 4; ++q) pm = smask[e][4 * jp + q] > pm ? smask[e][4 * jp + q] : pm;
        spm[tid] = jp < J ? pm : 0.0f;
    }
    __syncthreads();
    for (int x = tid; x < epw * kKD; x += 256) {  // U0, U1 of (entity, channel d): pooled positions in order
        const int e = x / kKD, d = x % kKD;
        float u0 = 0.0f, u1 = 0.0f;
        for (int jp = 0; jp < J; ++jp) {
            const float pm = spm[e * kJMax + jp];
            u0 = u0 + pm * ts[(e * kJMax + jp) * TS + d];
            if (jp + 1 < J) u1 = u1 + pm * ts[(e * kJMax + jp + 1) * TS + d];
        }
        us[(e * 2) * kKD + d] = u0;
        us[(e * 2 + 1) * kKD + d] = u1;
    }
    if (tid < 4) {
        float s = 0.0f;
        for (int jp = 0; jp < J; ++jp) s = s + spm[tid * kJMax + jp];
        spm[4 * kJMax + tid] = s;
    }
    __syncthreads();
    // wave e finishes local entity e: lane l the channels l and l + 64
    if (wave < epw) {
        const int64_t ie = (int64_t)blockIdx.x * epw + wave;
        const float S = spm[4 * kJMax + wave];
        const float* u0 = us + (wave * 2) * kKD;
        const float* u1 = u0 + kKD;
        float v[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int c = lane + 64 * h;
            const float* wrow = w2 + (int64_t)c * kKD * 2;
            float a = b2[c] * S;
            for (int d = 0; d < kKD; d += 2) {
                const float4 w = *reinterpret_cast<const float4*>(wrow + 2 * d);  // W2[c][d][0], [c][d][1], [c][d + 1][0], [c][d + 1][1]
                a = a + w.x * u0[d];
                a = a + w.y * u1[d];
                a = a + w.z * u0[d + 1];
                a = a + w.w * u1[d + 1];
            }
            v[h] = tanhf(a / S);
        }
        float scale = 1.0f;
        if (normalize) {  // F.normalize: x / max(||x||_2, 1e-12)
            float ss = v[0] * v[0] + v[1] * v[1];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off);
            const float nrm = sqrtf(ss);
            scale = nrm > 1e-12f ? nrm : 1e-12f;  // (divided by, as F.normalize does)
        }
        if (ie < n) {
            out[ie * ldo + lane] = normalize ? v[0] / scale : v[0];
            out[ie * ldo + lane + 64] = normalize ? v[1] / scale : v[1];
        }
    }
    if (bad && i_real < n) atomicMin(bad_tok, -1);
}

bool dkrl_rows_supported(int E, int D, int L) { return E > 0 && E % 4 == 0 && D == kKD && L >= 4 && L <= kKMaxL; }

template <int NSPLIT>
static hipError_t launch_dkrl_split(const int64_t* tok, const float* mask, int64_t n, int L, const float* emb, int64_t V, int E,
                                    const float* w1, const float* b1, const float* w2, const float* b2, int normalize, float* out,
                                    int64_t ldo, int* bad_tok, hipStream_t stream) {
    const int epw = (4 / NSPLIT) / ((L + 31) / 32);
    const int64_t blocks = (n + epw - 1) / epw;
    if (blocks > 0x7fffffff) return hipErrorInvalidValue;
    const size_t stage = (size_t)2 * kKStage * 4, epi = (size_t)(4 * (kKMaxL / 4) * (kKD + 1) + 4 * 2 * kKD + 4 * (kKMaxL / 4) + 4) * 4;
    const size_t lds = stage > epi ? stage : epi;
    static_assert(2 * kKStage * 4 > 64 * 1024, "the stages need the dynamic-LDS limit raised");
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&dkrl_rows_kernel<NSPLIT>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    dkrl_rows_kernel<NSPLIT><<<dim3((unsigned)blocks), 256, lds, stream>>>(tok, mask, n, L, emb, V, E, w1, b1, w2, b2, normalize, out, ldo,
                                                                          bad_tok);
    return hipGetLastError();
}

hipError_t launch_dkrl_rows(const int64_t* tok, const float* mask, int64_t n, int L, const float* emb, int64_t V, int E,
                            const float* w1, const float* b1, const float* w2, const float* b2, int normalize, float* out,
                            int64_t ldo, int* bad_tok, int n_cu, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    // M-tiles of the chunk: with fewer than four per CU the tiles are split across waves (2, then 4 waves per tile; an entity of
    // more than 32 tokens is two tiles: at most 2 waves per tile, so that it stays in one workgroup)
    const int tiles = (L + 31) / 32;
    const int64_t m_tiles = n * tiles;
    const long long forced = knob(KNOB_DKRL_SPLIT);
    // ([measured, tools/dkrl_probe.py] 512 entities x 32 tokens, E = 768, us per chunk at 1 / 2 / 4 waves per tile: 135 / 83 / 110)
    int split = m_tiles >= 4ll * n_cu ? 1 : (m_tiles >= n_cu || tiles == 2 ? 2 : 4);
    if (forced == 1 || forced == 2 || (forced == 4 && tiles == 1)) split = (int)forced;
    if (split == 1) return launch_dkrl_split<1>(tok, mask, n, L, emb, V, E, w1, b1, w2, b2, normalize, out, ldo, bad_tok, stream);
    if (split == 2) return launch_dkrl_split<2>(tok, mask, n, L, emb, V, E, w1, b1, w2, b2, normalize, out, ldo, bad_tok, stream);
    return launch_dkrl_split<4>(tok, mask, n, L, emb, V, E, w1, b1, w2, b2, normalize, out, ldo, bad_tok, stream);
}

}  // namespace blp
