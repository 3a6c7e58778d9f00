// rank_mfma.hip -- all-entities ranking for LARGE query blocks on the f32 matrix cores, bit-exact.
//
// Why the matrix pipe for an L1 distance.  The reference's TransE score is a strictly sequential f32
// sum over d of |x_d|, x_d = (h_d + r_d) - t_d, so every (candidate, query) pair owns one dependent
// chain of adds -- that part is VALU work by nature.  But producing x_d for a 32 x 32 block of
// (candidate, query) pairs is an outer "sum": x[c][q] = coef_q[d] - e_c[d], and
// v_mfma_f32_32x32x2_f32 is, bit for bit, the k-ordered chain  D = fma(A1, B1, fma(A0, B0, C))  in
// plain f32 (MI355X guide, section 3; re-verified on hardware for subnormals, signed zeros and
// arbitrary bit patterns with tools/mfma_exact_check.hip: only the sign of a zero result can differ).
// With  A = (e_c[d], 1),  B = (-1, coef_q[d]),  C = 0  it returns  RN(coef_q[d] - e_c[d])  for all 1024
// pairs: the matrix core does the operand broadcast AND the subtraction with the reference's single
// rounding, and the VALU is left with exactly one instruction per pair and element (acc += |x|).
//   B = (+1, r_q[d])  gives  RN(e + r)   (head-replacing TransE queries; x = that - t_q, one more VALU op)
// The lane-per-candidate VALU kernel (rank_all.hip) needs a wave-uniform coefficient operand per
// instruction, and on gfx950 every way of providing one stalls or runs at a fraction of the VALU rate
// (scalar loads return out of order; DPP operands issue at ~1/3 rate; LDS broadcast reads saturate the
// LDS).  Here each operand value is used by 32 partners inside the MFMA, operand traffic is tiny,
// registers are few, and the limit becomes the matrix pipe: 64 cycles per 1024 pair-elements.
//
// Layout.  A wave owns a 32-row candidate tile as the MFMA A operand, one VGPR per element d
// (lanes 0-31: e_c[d]; lanes 32-63: the constant 1 of the k = 1 slot).  Queries come in tiles of 32
// (MFMA columns).  The prep kernel writes every query tile's coefficients straight into the "B image"
// the kernel consumes -- [coefficient quad g][33 slots][4 floats]: slot j < 32 holds coefficients
// 4g..4g+3 of query j, slot 32 the k = 0 constant (-1 or +1) -- so staging a tile is a linear LDS-DMA
// copy, the 32 data lanes read 512 contiguous bytes per ds_read_b128 (conflict-free) and the other 32
// lanes broadcast-read the constant slot.  Workgroup = 4 waves = 4 candidate tiles sharing the staged
// query tiles (double-buffered); the accumulator tile D[c][q] puts queries on lanes, so the rank
// counts of a query are a per-lane sum over the 16 accumulator registers, combined across the two
// half-waves and the workgroup's waves with LDS atomics and flushed once with 64-bit global atomics.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "launch.h"
#include "rank_common.h"
#include "score_core.h"

#pragma clang fp contract(off)

namespace blp {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kMW = 4;              // waves per workgroup (one candidate tile each)
constexpr int kQT = 32;             // queries per tile (MFMA columns)
constexpr int kCT = 32;             // candidates per tile (MFMA rows)
constexpr int kTilesPerChunk = 8;   // query tiles per workgroup
constexpr int kSlots = 33;          // float4 slots per coefficient quad: 32 queries + 1 constant
constexpr int kSlabStride = 36;     // dwords, candidate-tile transpose slab (see rank_all.hip)

template <int MODEL, int SIDE>
struct MfmaSide;  // kConst: the k = 0 constant of the B operand
template <> struct MfmaSide<TRANSE, TAIL> { static constexpr float kConst = -1.0f; };
template <> struct MfmaSide<TRANSE, HEAD> { static constexpr float kConst = 1.0f; };

template <int C>
__host__ __device__ constexpr int tile_f4() { return (C / 4) * kSlots; }  // float4 per query-tile image

// Coefficient I of the query in slot j of a tile image (p = image + 4 * j floats).
struct ImgCoef {
    const float* __restrict__ p;
    template <int I>
    __device__ __forceinline__ float operator()(ic<I>) const { return p[(I >> 2) * (4 * kSlots) + (I & 3)]; }
};

// ------------------------------------------------------------------------------------------------
template <int MODEL, int SIDE, int D>
__device__ __forceinline__ void write_image_f4(const float* __restrict__ q_fixed, const float* __restrict__ q_rel,
                                               int64_t q0, int64_t n_side, int64_t idx, float4* __restrict__ img) {
    using S = Scorer<MODEL, SIDE, D>;
    constexpr int F4 = tile_f4<S::C>();
    const int64_t tile = idx / F4;
    const int rem = (int)(idx % F4), g = rem / kSlots, slot = rem % kSlots;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (slot == 32) {
        const float k = MfmaSide<MODEL, SIDE>::kConst;
        v = make_float4(k, k, k, k);
    } else {
        const int64_t q = tile * kQT + slot;
        if (q < n_side) {
            const float* f = q_fixed + (q0 + q) * D;
            const float* r = q_rel + (q0 + q) * D;
            v = make_float4(S::coef(f, r, 4 * g), S::coef(f, r, 4 * g + 1), S::coef(f, r, 4 * g + 2), S::coef(f, r, 4 * g + 3));
        }
    }
    img[idx] = v;
}

template <int MODEL, int D>
__global__ void prep_image_kernel(const float* __restrict__ q_fixed, const float* __restrict__ q_rel,
                                  int64_t q_head, int64_t q_tail, float4* __restrict__ img_head,
                                  float4* __restrict__ img_tail) {
    const int64_t n_h = (q_head + kQT - 1) / kQT * tile_f4<Scorer<MODEL, HEAD, D>::C>();
    const int64_t n_t = (q_tail + kQT - 1) / kQT * tile_f4<Scorer<MODEL, TAIL, D>::C>();
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n_h + n_t; i += (int64_t)gridDim.x * blockDim.x) {
        if (i < n_h) write_image_f4<MODEL, HEAD, D>(q_fixed, q_rel, 0, q_head, i, img_head);
        else write_image_f4<MODEL, TAIL, D>(q_fixed, q_rel, q_head, q_tail, i - n_h, img_tail);
    }
}

template <int MODEL, int SIDE, int D>
__device__ __forceinline__ float score_from_image(const float (&e)[D], const float4* img, int64_t q_local) {
    using S = Scorer<MODEL, SIDE, D>;
    const float* p = reinterpret_cast<const float*>(img + (q_local / kQT) * tile_f4<S::C>()) + 4 * (q_local % kQT);
    return S::template score<false>(e, ImgCoef{p});
}

template <int MODEL, int D>
__global__ __launch_bounds__(64) void true_key_img_kernel(const float* __restrict__ table, int64_t ld,
                                                         const int64_t* __restrict__ true_row,
                                                         const float* __restrict__ q_true,
                                                         const float4* __restrict__ img_head,
                                                         const float4* __restrict__ img_tail, int64_t q_head,
                                                         int64_t q_tail, float* __restrict__ key_true) {
    const int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (q >= q_head + q_tail) return;
    float e[D];
    load_row<D>(e, true_row ? table + true_row[q] * ld : q_true + q * D);
    key_true[q] = q < q_head ? score_from_image<MODEL, HEAD, D>(e, img_head, q)
                             : score_from_image<MODEL, TAIL, D>(e, img_tail, q - q_head);
}

template <int MODEL, int D>
__global__ __launch_bounds__(256) void filt_counts_img_kernel(
    const float* __restrict__ table, int64_t ld, const float4* __restrict__ img_head,
    const float4* __restrict__ img_tail, const float* __restrict__ key_true, int64_t q_head, int64_t q_tail,
    const int64_t* __restrict__ rowptr, const int64_t* __restrict__ col, unsigned long long* __restrict__ acc_f) {
    const int64_t q = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (q >= q_head + q_tail) return;
    const int64_t lo = rowptr[q], hi = rowptr[q + 1];
    const float kt = key_true[q];
    unsigned gt = 0, ge = 0;
    for (int64_t k = lo + lane; k < hi; k += 64) {
        float e[D];
        load_row<D>(e, table + col[k] * ld);
        const float key = q < q_head ? score_from_image<MODEL, HEAD, D>(e, img_head, q)
                                     : score_from_image<MODEL, TAIL, D>(e, img_tail, q - q_head);
        gt += key > kt;
        ge += key >= kt;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        gt += __shfl_down(gt, off);
        ge += __shfl_down(ge, off);
    }
    if (lane == 0) acc_f[q] = (unsigned long long)gt | ((unsigned long long)ge << 32);
}

// ------------------------------------------------------------------------------------------------
// 32 candidate rows -> the MFMA A operand: a[d] = e_{row0 + lane}[d] for lanes 0-31, 1.0 for lanes 32-63.
// Coalesced loads (8 rows x 128 B per wave instruction), transposed through a wave-private LDS slab.
template <int D>
__device__ __forceinline__ void load_a_tile(float (&a)[D], const float* __restrict__ table, int64_t N, int64_t ld,
                                            int64_t row0, float* slab, int lane) {
    const int sub_row = lane >> 3, sub_col = (lane & 7) * 4;
    static_for<4>([&](auto ii) {
        constexpr int i = decltype(ii)::value;
        int64_t row = row0 + 8 * i + sub_row;
        row = row < N ? row : N - 1;
        const float* src = table + row * ld + sub_col;
        static_for<D / 32>([&](auto ss) {
            constexpr int s = decltype(ss)::value;
            const float4 v = *reinterpret_cast<const float4*>(src + s * 32);
            a[32 * s + 4 * i] = v.x; a[32 * s + 4 * i + 1] = v.y;
            a[32 * s + 4 * i + 2] = v.z; a[32 * s + 4 * i + 3] = v.w;
        });
    });
    float* wr = slab + sub_row * kSlabStride + sub_col;
    const float* rd = slab + (lane & 31) * kSlabStride;
    static_for<D / 32>([&](auto ss) {
        constexpr int s = decltype(ss)::value;
        if (s > 0) wave_lds_sync();
        static_for<4>([&](auto ii) {
            constexpr int i = decltype(ii)::value;
            *reinterpret_cast<float4*>(wr + 8 * i * kSlabStride) =
                make_float4(a[32 * s + 4 * i], a[32 * s + 4 * i + 1], a[32 * s + 4 * i + 2], a[32 * s + 4 * i + 3]);
        });
        wave_lds_sync();
        static_for<8>([&](auto jj) {
            constexpr int j = decltype(jj)::value;
            const float4 w = *reinterpret_cast<const float4*>(rd + 4 * j);
            a[32 * s + 4 * j] = w.x; a[32 * s + 4 * j + 1] = w.y;
            a[32 * s + 4 * j + 2] = w.z; a[32 * s + 4 * j + 3] = w.w;
        });
    });
    if (lane >= 32) static_for<D>([&](auto d) { a[d] = 1.0f; });
}

// LDS-DMA copy of one query-tile image (BYTES, a multiple of 16) by the whole workgroup.
template <int BYTES>
__device__ __forceinline__ void stage_tile(const float4* __restrict__ src, float* dst, int wave, int lane) {
    constexpr int ROUNDS = (BYTES + kMW * 1024 - 1) / (kMW * 1024);
    static_for<ROUNDS>([&](auto kk) {
        constexpr int k = decltype(kk)::value;
        const int seg = (k * kMW + wave) * 1024;  // byte offset of this wave instruction
        if (seg + lane * 16 < BYTES)
            __builtin_amdgcn_global_load_lds((global_cptr)(reinterpret_cast<const char*>(src) + seg + lane * 16),
                                             (lds_ptr)(reinterpret_cast<char*>(dst) + seg), 16, 0, 0);
    });
}

// One query tile against the wave's candidate tile: D MFMAs + the sequential |x| chains; returns this
// lane's rank counts for query (lane & 31) over its 16 candidate rows.
template <int MODEL, int SIDE, int D>
__device__ __forceinline__ void tile_pass(const float (&a)[D], const float* buf, float kt, unsigned row_mask,
                                          int lane, unsigned& gt, unsigned& ge) {
    static_assert(MODEL == TRANSE, "bilinear MFMA plans are not written yet");
    const float4* b_ptr = reinterpret_cast<const float4*>(buf) + (lane < 32 ? 32 : lane - 32);
    const float4* t_ptr = reinterpret_cast<const float4*>(buf) + (D / 4) * kSlots + (lane & 31);
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    f32x16 acc = zero;
    // One MFMA per element.  Software pipeline, pinned with sched_barrier (left alone, the scheduler
    // hoists all D independent MFMAs to the top and spills their results): the MFMA of element d + 1
    // is issued, then the VALU consumes element d from the other result buffer; the coefficient quads
    // of the next 4 elements are read from LDS one group ahead.
    auto comp = [](const float4& v, auto ss) {
        constexpr int s = decltype(ss)::value;
        return s == 0 ? v.x : s == 1 ? v.y : s == 2 ? v.z : v.w;
    };
    // acc[r] += |x[r]| as one v_add_f32 with the |.| source modifier per element, written in inline
    // asm: the C++ vector add is legalised to v_pk_add_f32 (no |x| modifier -> an extra v_and per
    // element, and packed f32 is no faster than two plain adds on gfx950), and 16 scalar C++ adds make
    // the register allocator spill the MFMA result tuples.  hipcc pads no hazards for an asm consumer,
    // so the tail-side sequence (whose first reader of the MFMA result is the asm) carries its own
    // s_nop: MFMA(d+1) issue + 16 adds + s_nop 4 >= the 19 wait states a 16-pass MFMA result needs
    // (LLVM GCNHazardRecognizer, gfx950).  The head side reads the result first with compiler-visible
    // v_sub instructions, which the hazard recogniser pads itself.
    auto consume = [&](auto dd, const f32x16& x, float t) {
        constexpr int d = decltype(dd)::value;
        f32x16 v = x;
        if constexpr (SIDE == HEAD) v = v - t;  // (e + r) - t, models.py:223
        static_for<16>([&](auto rr) {
            constexpr int r = decltype(rr)::value;
            float ar = acc[r];
            const float vr = v[r];
            if constexpr (d == 0) {
                if constexpr (r == 0 && SIDE == TAIL) asm volatile("s_nop 4\n\tv_and_b32 %0, 0x7fffffff, %1" : "=v"(ar) : "v"(vr));
                else asm volatile("v_and_b32 %0, 0x7fffffff, %1" : "=v"(ar) : "v"(vr));
            } else {
                if constexpr (r == 0 && SIDE == TAIL) asm volatile("s_nop 4\n\tv_add_f32_e64 %0, %0, |%1|" : "+v"(ar) : "v"(vr));
                else asm volatile("v_add_f32_e64 %0, %0, |%1|" : "+v"(ar) : "v"(vr));
            }
            acc[r] = ar;
        });
    };
    float4 b_cur = b_ptr[0], t_cur = make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr (SIDE == HEAD) t_cur = t_ptr[0];
    f32x16 x0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], b_cur.x, zero, 0, 0, 0), x1 = zero;
    static_for<D / 4>([&](auto gg) {
        constexpr int g = decltype(gg)::value;
        float4 b_next = b_cur, t_next = t_cur;
        if constexpr (g + 1 < D / 4) {
            b_next = b_ptr[(g + 1) * kSlots];
            if constexpr (SIDE == HEAD) t_next = t_ptr[(g + 1) * kSlots];
        }
        static_for<4>([&](auto ss) {
            constexpr int s = decltype(ss)::value, d = 4 * g + s;
            f32x16& xin = (d & 1) ? x1 : x0;   // holds element d
            f32x16& xout = (d & 1) ? x0 : x1;  // receives element d + 1
            if constexpr (d + 1 < D) {
                const float b = s < 3 ? comp(b_cur, ic<(s + 1) & 3>{}) : b_next.x;
                xout = __builtin_amdgcn_mfma_f32_32x32x2f32(a[d + 1], b, zero, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            consume(ic<d>{}, xin, comp(t_cur, ss));
            __builtin_amdgcn_sched_barrier(0);
        });
        b_cur = b_next;
        t_cur = t_next;
    });
    static_for<16>([&](auto rr) {
        constexpr int r = decltype(rr)::value;
        const float key = -acc[r];
        const bool ok = (row_mask >> r) & 1u;
        gt += ok && key > kt;
        ge += ok && key >= kt;
    });
}

template <int MODEL, int D>
__global__ __launch_bounds__(kMW * 64, 2) void rank_mfma_kernel(
    const float* __restrict__ table, int64_t N, int64_t ld, const float4* __restrict__ img_head,
    const float4* __restrict__ img_tail, const float* __restrict__ key_true, int q_head, int q_tail,
    int n_quads, int chunks_head, unsigned long long* __restrict__ acc) {
    constexpr int CH = Scorer<MODEL, HEAD, D>::C, CT = Scorer<MODEL, TAIL, D>::C;
    constexpr int BUF_FLOATS = 4 * tile_f4<(CH > CT ? CH : CT)>();
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* buf0 = smem;
    float* buf1 = smem + BUF_FLOATS;
    unsigned* cnt = reinterpret_cast<unsigned*>(smem + 2 * BUF_FLOATS);  // [kTilesPerChunk * 32][2]
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;

    const int quad = blockIdx.x % n_quads;
    const int chunk = blockIdx.x / n_quads;
    const bool head = chunk < chunks_head;
    const int side_chunk = head ? chunk : chunk - chunks_head;
    const int n_side = head ? q_head : q_tail;
    const int tile0 = side_chunk * kTilesPerChunk;
    const int n_side_tiles = (n_side + kQT - 1) / kQT;
    const int n_tiles = n_side_tiles - tile0 < kTilesPerChunk ? n_side_tiles - tile0 : kTilesPerChunk;
    const float* kt_side = key_true + (head ? 0 : q_head);

    // candidate tile -> A operand (the slabs alias the query buffers: nothing is staged yet)
    float a[D];
    const int64_t row0 = ((int64_t)quad * kMW + wave) * kCT;
    load_a_tile<D>(a, table, N, ld, row0, smem + wave * (kCT * kSlabStride), lane);
    unsigned row_mask = 0;  // bit r: accumulator register r of this lane is a real table row
#pragma unroll
    for (int r = 0; r < 16; ++r)
        row_mask |= (unsigned)(row0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5) < N) << r;
    for (int i = tid; i < kTilesPerChunk * kQT * 2; i += kMW * 64) cnt[i] = 0;
    __syncthreads();

    auto run = [&](auto side_tag) {
        constexpr int SIDE = decltype(side_tag)::value;
        constexpr int C = Scorer<MODEL, SIDE, D>::C;
        constexpr int BYTES = tile_f4<C>() * 16;
        const float4* img = (SIDE == HEAD ? img_head : img_tail) + (int64_t)tile0 * tile_f4<C>();
        stage_tile<BYTES>(img, buf0, wave, lane);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int t = 0; t < n_tiles; ++t) {
            float* cur = (t & 1) ? buf1 : buf0;
            if (t + 1 < n_tiles) stage_tile<BYTES>(img + (int64_t)(t + 1) * tile_f4<C>(), (t & 1) ? buf0 : buf1, wave, lane);
            const int q = (tile0 + t) * kQT + (lane & 31);
            const float kt = kt_side[q < n_side ? q : n_side - 1];
            unsigned gt = 0, ge = 0;
            tile_pass<MODEL, SIDE, D>(a, cur, kt, row_mask, lane, gt, ge);
            atomicAdd(cnt + 2 * (t * kQT + (lane & 31)), gt);
            atomicAdd(cnt + 2 * (t * kQT + (lane & 31)) + 1, ge);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    };
    if (head) run(ic<HEAD>{}); else run(ic<TAIL>{});

    for (int i = tid; i < n_tiles * kQT; i += kMW * 64) {
        const int q = tile0 * kQT + i;
        const unsigned long long v = (unsigned long long)cnt[2 * i] | ((unsigned long long)cnt[2 * i + 1] << 32);
        if (q < n_side && v) atomicAdd(acc + (head ? 0 : q_head) + q, v);
    }
}

// ------------------------------------------------------------------------------------------------
struct MfmaWorkspace {
    float4* img_head;
    float4* img_tail;
    float* key_true;
    unsigned long long* acc;
    unsigned long long* acc_f;
    size_t bytes;
};

template <int MODEL, int D>
static MfmaWorkspace carve(void* base, int64_t q_head, int64_t q_tail) {
    MfmaWorkspace w;
    const int64_t Q = q_head + q_tail;
    char* p = static_cast<char*>(base);
    size_t off = 0;
    const size_t nh = (size_t)((q_head + kQT - 1) / kQT) * tile_f4<Scorer<MODEL, HEAD, D>::C>() * 16;
    const size_t nt = (size_t)((q_tail + kQT - 1) / kQT) * tile_f4<Scorer<MODEL, TAIL, D>::C>() * 16;
    w.img_head = reinterpret_cast<float4*>(p + off); off = align_up(off + nh, 256);
    w.img_tail = reinterpret_cast<float4*>(p + off); off = align_up(off + nt, 256);
    w.key_true = reinterpret_cast<float*>(p + off);  off = align_up(off + (size_t)Q * 4, 256);
    w.acc = reinterpret_cast<unsigned long long*>(p + off);   off = align_up(off + (size_t)Q * 8, 256);
    w.acc_f = reinterpret_cast<unsigned long long*>(p + off); off = align_up(off + (size_t)Q * 8, 256);
    w.bytes = off;
    return w;
}

bool rank_mfma_applicable(int model, int D, int64_t q_head, int64_t q_tail) {
    // The 32-query MFMA tile pays off once tiles are reasonably full; small query blocks (and the
    // HBM-bound few-query case) stay on the lane-per-candidate kernel.
    return model == TRANSE && (D == 64 || D == 128) && q_head + q_tail >= 64;
}

size_t rank_mfma_workspace_bytes(int model, int D, int64_t q_head, int64_t q_tail) {
    if (model != TRANSE) return 0;
    if (D == 64) return carve<TRANSE, 64>(nullptr, q_head, q_tail).bytes;
    if (D == 128) return carve<TRANSE, 128>(nullptr, q_head, q_tail).bytes;
    return 0;
}

template <int MODEL, int D>
static hipError_t rank_mfma_impl(const float* table, int64_t N, int64_t ld, const float* q_fixed,
                                 const float* q_rel, const int64_t* true_row, const float* q_true,
                                 int64_t q_head, int64_t q_tail, const int64_t* filt_rowptr,
                                 const int64_t* filt_col, int32_t* counts, void* workspace, hipStream_t stream,
                                 hipEvent_t ev_start, hipEvent_t ev_stop) {
    const int64_t Q = q_head + q_tail;
    MfmaWorkspace w = carve<MODEL, D>(workspace, q_head, q_tail);
    hipError_t err = hipMemsetAsync(w.acc, 0, (size_t)Q * 8, stream);
    if (err != hipSuccess) return err;
    constexpr int CH = Scorer<MODEL, HEAD, D>::C, CT = Scorer<MODEL, TAIL, D>::C;
    const int64_t th = (q_head + kQT - 1) / kQT, tt = (q_tail + kQT - 1) / kQT;
    {
        const int64_t total = th * tile_f4<CH>() + tt * tile_f4<CT>();
        const int64_t blocks = (total + 255) / 256;
        prep_image_kernel<MODEL, D><<<(int)(blocks < 16384 ? blocks : 16384), 256, 0, stream>>>(
            q_fixed, q_rel, q_head, q_tail, w.img_head, w.img_tail);
        true_key_img_kernel<MODEL, D><<<(int)((Q + 63) / 64), 64, 0, stream>>>(
            table, ld, true_row, q_true, w.img_head, w.img_tail, q_head, q_tail, w.key_true);
    }
    if (N > 0) {
        const int64_t n_ctiles = (N + kCT - 1) / kCT;
        const int64_t n_quads = (n_ctiles + kMW - 1) / kMW;
        const int64_t chunks_head = (th + kTilesPerChunk - 1) / kTilesPerChunk;
        const int64_t chunks_tail = (tt + kTilesPerChunk - 1) / kTilesPerChunk;
        const int64_t blocks = n_quads * (chunks_head + chunks_tail);
        constexpr int BUF_FLOATS = 4 * tile_f4<(CH > CT ? CH : CT)>();
        const size_t lds = (size_t)2 * BUF_FLOATS * 4 + (size_t)kTilesPerChunk * kQT * 2 * 4;
        static_assert(kMW * kCT * kSlabStride <= 2 * BUF_FLOATS, "transpose slabs must fit the query buffers");
        if (ev_start) (void)hipEventRecord(ev_start, stream);
        rank_mfma_kernel<MODEL, D><<<dim3((unsigned)blocks), kMW * 64, lds, stream>>>(
            table, N, ld, w.img_head, w.img_tail, w.key_true, (int)q_head, (int)q_tail, (int)n_quads,
            (int)chunks_head, w.acc);
        if (ev_stop) (void)hipEventRecord(ev_stop, stream);
    }
    const bool filtered = filt_rowptr != nullptr;
    if (filtered)
        filt_counts_img_kernel<MODEL, D><<<(int)((Q + 3) / 4), 256, 0, stream>>>(
            table, ld, w.img_head, w.img_tail, w.key_true, q_head, q_tail, filt_rowptr, filt_col, w.acc_f);
    err = launch_finalize_counts(w.acc, filtered ? w.acc_f : nullptr, Q, counts, stream);
    return err != hipSuccess ? err : hipGetLastError();
}

hipError_t launch_rank_all_mfma(int model, int D, const float* table, int64_t N, int64_t ld,
                                const float* q_fixed, const float* q_rel, const int64_t* true_row,
                                const float* q_true, int64_t q_head, int64_t q_tail,
                                const int64_t* filt_rowptr, const int64_t* filt_col, int32_t* counts,
                                void* workspace, int n_cu, hipStream_t stream, hipEvent_t ev_start,
                                hipEvent_t ev_stop) {
    (void)n_cu;
    if (model == TRANSE && D == 128)
        return rank_mfma_impl<TRANSE, 128>(table, N, ld, q_fixed, q_rel, true_row, q_true, q_head, q_tail,
                                           filt_rowptr, filt_col, counts, workspace, stream, ev_start, ev_stop);
    if (model == TRANSE && D == 64)
        return rank_mfma_impl<TRANSE, 64>(table, N, ld, q_fixed, q_rel, true_row, q_true, q_head, q_tail,
                                          filt_rowptr, filt_col, counts, workspace, stream, ev_start, ev_stop);
    return hipErrorInvalidValue;
}

}  // namespace blp
