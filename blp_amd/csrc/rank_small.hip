// rank_small.hip -- exact f32 ranking of a SMALL query block against the whole table: the reference's own
// evaluation batches (train.py:128-171: 64 triples = 128 queries against 14 541 entities, scripts/*.sh eval_batch_size
// 16 .. 128), where a call is a handful of microseconds of arithmetic and everything else is latency.
//
// As in rank_tiles (rank_all.hip) a lane owns one table row in D VGPRs and every (candidate, query) key is computed by
// the Scorer<> instruction sequence (score_core.h), so keys -- and counts -- are bit-identical to the other exact
// kernels'.  What differs is what a small block is bound by (measured, tools/small_timing.py):
//   * a wave issues at most one VALU instruction per ~4.5 cycles however independent its instructions are; two waves per
//     SIMD are needed to reach the pipe's 2.4.  A 128-query block must therefore be cut into about 2 048 equal waves
//     -- two per SIMD, all resident at once -- and into no more, because
//   * every wave that holds a tile has to fetch it: with one tile per wave and 16 queries per workgroup the 7.4 MB
//     FB15k-237 table was read 8 times per call, 10 us of a 35 us kernel.  Here the FOUR waves of a workgroup share ONE
//     tile of 64 rows (fetched once, coalesced, handed to every wave through LDS) and split the workgroup's queries, so
//     a (tile, 64-query chunk) workgroup reads 32 KB for 64 queries;
//   * no coefficient array and no prep launch: the workgroup computes the coefficients of 32 queries at a time
//     (Scorer<>::coef on q_fixed / q_rel) into LDS, the first batch under the tile's global loads;
//   * the coefficients are read back as LDS broadcasts (ds_read_b128, every lane the same address: four coefficients
//     per instruction, 4 LDS cycles, no conflicts) with hand-placed waits, because the compiler waits for an LDS
//     read right where it issues it and a lone wave then pays the latency on every step; TransE scores four (tail
//     side) or two (head side) queries together so that eight reads cover a step.
//   * no atomics: 228 workgroups adding to one query's accumulator serialise at the memory side (device-scope atomics
//     are not served by the XCD-local L2): +30 us.  Every (wave, query) pair is unique in a workgroup, so a workgroup
//     keeps plain counters in LDS and stores them as its slot's partial counts; the finalize kernel adds the slots.
// Launch chain of a small block: true_key -> rank_small -> filter + finalize.
// Two kernels: rank_small_kernel (all of the above; the bilinear models, and TransE against tables of more than 1 024
// tiles) and rank_small_sgpr_kernel (TransE: the tile stays in LDS, the coefficients come as scalar registers from rows the
// true-key launch materialises; further down).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "knobs.h"
#include "launch.h"
#include "rank_common.h"
#include "score_core.h"
#include "tile.h"

#pragma clang fp contract(off)

namespace blp {

constexpr int kSmallRound = 32;     // queries whose coefficients are in LDS at a time: 8 per wave
constexpr int kSmallPerWave = kSmallRound / kWaves;
constexpr int kSmallMaxChunk = 1024; // queries per workgroup (a counter each in LDS)
__host__ __device__ constexpr int small_coef_stride(int D) { return 2 * D + 4; }  // floats per query in LDS

template <int K>
__device__ __forceinline__ float comp4(const float4& v) { return K == 0 ? v.x : K == 1 ? v.y : K == 2 ? v.z : v.w; }

// A query's coefficient row in LDS, read as broadcast float4s (the compiler merges the four uses of one quad).
struct LdsCoef {
    const float* p;
    template <int I>
    __device__ __forceinline__ float operator()(ic<I>) const {
        const float4 v = *reinterpret_cast<const float4*>(p + (I & ~3));
        return comp4<I & 3>(v);
    }
};

// ---- hand-pipelined LDS broadcasts -----------------------------------------------------------------------------
// The compiler places an LDS read next to its first use and waits for it there: with one or two waves per SIMD every
// step of the unrolled element loop then pays the LDS latency in full (24 us for 16 queries -- no better than the scalar
// loads of rank_tiles).  Here the reads are inline asm (the compiler does not track their completion), issued one
// step -- 4 elements of every query of the group -- ahead, and the only wait is a hand-placed s_waitcnt at the END of
// the step whose arithmetic covered them.  Register dependences keep the order: the wait asm takes the in-flight
// registers and the partial sums as read-write operands, so no use of a loaded value can move above it and the step's
// arithmetic cannot sink below it; an empty asm pinning the partial sums right after the reads keeps the step's
// arithmetic from being scheduled ahead of them.
typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned lds_address(const float* p) {
    return (unsigned)(size_t)(__attribute__((address_space(3))) const float*)p;
}
template <int OFF>
__device__ __forceinline__ f4 lds_bcast16(unsigned addr) {  // every lane the same address: 4 cycles, no conflicts
    f4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(OFF));
    return v;
}
__device__ __forceinline__ void lds_landed(f4& a, f4& b, f4& c, f4& d) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
}
__device__ __forceinline__ void lds_landed(f4& a, f4& b, f4& c, f4& d, float& s0, float& s1) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(s0), "+v"(s1));
}
__device__ __forceinline__ void lds_landed(f4& a, f4& b, f4& c, f4& d, float& s0, float& s1, float& s2, float& s3) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3));
}
// Nothing that uses these registers -- the landed coefficients of the current step and the partial sums -- may be
// scheduled before this point (i.e. before the next step's reads, issued just above it).
__device__ __forceinline__ void pin(f4& a, f4& b, f4& c, f4& d, float& s0, float& s1) {
    asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(s0), "+v"(s1));
}
__device__ __forceinline__ void pin(f4& a, f4& b, f4& c, f4& d, float& s0, float& s1, float& s2, float& s3) {
    asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3));
}

// TransE keys of G queries against the wave's tile: per query exactly Scorer<TRANSE, SIDE, D>::score<false>'s
// operations in its order -- (c - e) or ((e + r) - t), |.|, strict left-to-right sum -- the G chains interleaved.
// row[g]: LDS byte address of query g's coefficient row.  A step is 4 elements of every query of the group: tail side
// G = 4 queries (32 VALU instructions per step), head side -- two coefficient rows per query -- G = 2 (24): four reads
// in flight either way, covered by a lone wave's ~4.5 cycles per instruction.
template <int SIDE>
constexpr int transe_group() { return SIDE == TAIL ? 4 : 2; }

template <int SIDE, int D, int G>
__device__ __forceinline__ void transe_keys(const float (&e)[D], const unsigned (&row)[G], float (&key)[G]) {
    static_assert(G == transe_group<SIDE>() && D % 4 == 0, "step layout");
    constexpr int NS = D / 4;
    float acc[G] = {};
    f4 buf[2][4];  // [parity][slot]: tail slot = query g; head slot = 2g + w (w: 0 = r, 1 = t)
    auto issue = [&](auto ss) {
        constexpr int s = decltype(ss)::value;
        static_for<4>([&](auto tt) {
            constexpr int t = decltype(tt)::value;
            if constexpr (SIDE == TAIL) buf[s & 1][t] = lds_bcast16<16 * s>(row[t]);
            else buf[s & 1][t] = lds_bcast16<((t & 1) * D + 4 * s) * 4>(row[t >> 1]);
        });
    };
    issue(ic<0>{});
    lds_landed(buf[0][0], buf[0][1], buf[0][2], buf[0][3]);
    static_for<NS>([&](auto ss) {
        constexpr int s = decltype(ss)::value, p = s & 1;
        if constexpr (s + 1 < NS) {  // next step's reads first; the pin keeps this step's arithmetic behind them
            issue(ic<s + 1>{});
            if constexpr (G == 4) pin(buf[p][0], buf[p][1], buf[p][2], buf[p][3], acc[0], acc[1], acc[2], acc[3]);
            else pin(buf[p][0], buf[p][1], buf[p][2], buf[p][3], acc[0], acc[1]);
        }
        static_for<4>([&](auto kk) {
            constexpr int k = decltype(kk)::value, d = 4 * s + k;
            static_for<G>([&](auto gg) {
                constexpr int g = decltype(gg)::value;
                float x;
                if constexpr (SIDE == TAIL) {
                    x = buf[p][g][k] - e[d];
                } else {
                    const float y = e[d] + buf[p][2 * g][k];
                    x = y - buf[p][2 * g + 1][k];
                }
                const float cur = fabsf(x);
                acc[g] = d == 0 ? cur : acc[g] + cur;
            });
        });
        if constexpr (s + 1 < NS) {
            constexpr int n = (s + 1) & 1;
            if constexpr (G == 4) lds_landed(buf[n][0], buf[n][1], buf[n][2], buf[n][3], acc[0], acc[1], acc[2], acc[3]);
            else lds_landed(buf[n][0], buf[n][1], buf[n][2], buf[n][3], acc[0], acc[1]);
        }
    });
    static_for<G>([&](auto gg) { key[decltype(gg)::value] = -acc[decltype(gg)::value]; });
}

__device__ __forceinline__ void count_key(float key, float kt, bool valid, unsigned long long* cnt_q, int lane) {
    const unsigned long long gt = __popcll(__ballot(valid && key > kt));
    const unsigned long long ge = __popcll(__ballot(valid && key >= kt));
    if (lane == 0 && ge) atomicAdd(cnt_q, gt | (ge << 32));  // ds_add_u64, no return: nothing to wait for
}

// n queries of one side: coefficient rows coef[0 .. n) (stride small_coef_stride(D)) and true keys kt[0 .. n) in LDS,
// the wave's counters cnt[0 .. n) in LDS
template <int MODEL, int SIDE, int D>
__device__ __forceinline__ void score_side(const float (&e)[D], bool valid, const float* coef, const float* kt,
                                           unsigned long long* cnt, int n, int lane) {
    constexpr int CS = small_coef_stride(D);
    if constexpr (MODEL == TRANSE) {
        constexpr int G = transe_group<SIDE>();
        for (int j = 0; j < n; j += G) {
            unsigned row[G];
            static_for<G>([&](auto gg) {
                constexpr int g = decltype(gg)::value;
                row[g] = lds_address(coef + (j + g < n ? j + g : n - 1) * CS);  // the last group may repeat its last query
            });
            float key[G];
            transe_keys<SIDE, D, G>(e, row, key);
            static_for<G>([&](auto gg) {
                constexpr int g = decltype(gg)::value;
                if (j + g < n) count_key(key[g], kt[j + g], valid, cnt + j + g, lane);
            });
        }
    } else {  // the torch.sum order keeps 32 independent accumulators per key: one query at a time
        for (int j = 0; j < n; ++j) {
            const float key = Scorer<MODEL, SIDE, D>::template score<false>(e, LdsCoef{coef + j * CS});
            count_key(key, kt[j], valid, cnt + j, lane);
        }
    }
}

// -DBLP_TIMING: where a wave's time goes (100 MHz wall clock ticks summed over waves; tools/exact_small_probe.py)
#ifdef BLP_TIMING
__device__ unsigned long long g_small_timing[8];
#define BLP_ST(i) do { const unsigned long long now_ = wall_clock64(); tacc[i] += now_ - tlast; tlast = now_; } while (0)
#else
#define BLP_ST(i) do { } while (0)
#endif

// MULTI: a slot is several tiles (tables of more than kSmallMaxSlots tiles): the per-tile body in a loop, kept out of
// the one-tile kernel (loop-invariant code motion out of that loop costs registers the common case needs).
template <int MODEL, int D, bool MULTI>
__global__ __launch_bounds__(kWaves * 64, 2) void rank_small_kernel(
    const float* __restrict__ table, int64_t N, int64_t ld, const QRows q_fixed,
    const QRows q_rel, const float* __restrict__ key_true, int q_head, int q_tail, int n_tiles, int n_slots,
    int q_chunk, unsigned long long* __restrict__ partial) {
    using SH = Scorer<MODEL, HEAD, D>;
    using ST = Scorer<MODEL, TAIL, D>;
    constexpr int CS = small_coef_stride(D), CMAX = 2 * D, TS = D + 4;  // TS: row stride of the LDS tile, floats
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    float* tile = smem;                        // 64 x TS
    float* coef = tile + kTileRows * TS;       // kSmallRound x CS
    float* kt = coef + kSmallRound * CS;       // kSmallRound
    unsigned long long* cnt = reinterpret_cast<unsigned long long*>(kt + kSmallRound);  // kSmallMaxChunk

    // workgroup -> (query chunk, tile slot); consecutive workgroups share a query chunk.  A slot is one tile, or, for
    // a table of more than kSmallMaxSlots tiles, the tiles slot, slot + n_slots, ...
    const int chunk = blockIdx.x / n_slots, slot = blockIdx.x % n_slots;
    const int Q = q_head + q_tail;
    const int qa = chunk * q_chunk;
    const int qb = qa + q_chunk < Q ? qa + q_chunk : Q;
    for (int j = tid; j < qb - qa; j += kWaves * 64) cnt[j] = 0;

#ifdef BLP_TIMING
    unsigned long long tacc[4] = {0, 0, 0, 0}, tlast = wall_clock64();
#endif
    // Coefficients + true keys of queries [q0, q0 + nr) into LDS: thread <-> coefficient index, eight queries' worth of
    // loads in flight at a time (one load per loop iteration cost 32 L2 round trips: 5 us)
    auto stage = [&](int q0, int nr) {
        constexpr int QPP = kWaves * 64 / CMAX;  // queries per pass of the workgroup: 1 (D = 128) or 2 (D = 64)
        const int i = tid % CMAX, jo = tid / CMAX;
        for (int j0 = 0; j0 < nr; j0 += 8 * QPP) {
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int j = j0 + k * QPP + jo, q = q0 + (j < nr ? j : nr - 1);
                const float* f = q_fixed.row(q);
                const float* r = q_rel.row(q);
                v[k] = 0.0f;
                if (q < q_head) { if (i < SH::C) v[k] = SH::coef(f, r, i); }
                else            { if (i < ST::C) v[k] = ST::coef(f, r, i); }
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int j = j0 + k * QPP + jo;
                if (j < nr) coef[j * CS + i] = v[k];
            }
        }
        if (tid < nr) kt[tid] = key_true[q0 + tid];
    };
    const bool one_round = qb - qa <= kSmallRound;  // the coefficients then stay in LDS across the slot's tiles

    auto one_tile = [&](int t, bool first) {
        const int64_t row0 = (int64_t)t * kTileRows;
        // 1. the tile: thread t fetches 16-byte pieces t, t + 256, ... of the 64 x D block (whole 512-byte rows per 32
        //    threads); rows past the end of the table are clamped to the last row and masked when counting
        typedef float floatx4 __attribute__((ext_vector_type(4)));
        constexpr int kPieces = kTileRows * (D / 4) / (kWaves * 64);
        floatx4 piece[kPieces];
        static_for<kPieces>([&](auto kk) {
            constexpr int k = decltype(kk)::value;
            const int idx = tid + k * kWaves * 64, r = idx / (D / 4), c = idx % (D / 4);
            int64_t row = row0 + r;
            row = row < N ? row : N - 1;
            piece[k] = *reinterpret_cast<const floatx4*>(table + row * ld + 4 * c);
        });
        if (!first) __syncthreads();  // the previous tile has been taken out of LDS, its last round scored
        // 2. the first round's coefficients, under the tile's loads
        if (first || !one_round) stage(qa, qb - qa < kSmallRound ? qb - qa : kSmallRound);
        static_for<kPieces>([&](auto kk) {
            constexpr int k = decltype(kk)::value;
            const int idx = tid + k * kWaves * 64, r = idx / (D / 4), c = idx % (D / 4);
            *reinterpret_cast<floatx4*>(tile + r * TS + 4 * c) = piece[k];
        });
        BLP_ST(0);
        __syncthreads();
        // 3. every wave takes the whole tile, lane l <- row row0 + l (row stride D + 4 floats: conflict-free ds_read_b128)
        float e[D];
        static_for<D / 4>([&](auto jj) {
            constexpr int j = decltype(jj)::value;
            const float4 w = *reinterpret_cast<const float4*>(tile + lane * TS + 4 * j);
            e[4 * j] = w.x; e[4 * j + 1] = w.y; e[4 * j + 2] = w.z; e[4 * j + 3] = w.w;
        });
        const bool valid = row0 + lane < N;
        BLP_ST(1);

        // 4. rounds of kSmallRound queries, kSmallPerWave per wave: head-replacing queries first (train.py:149 order)
        for (int q0 = qa; q0 < qb; q0 += kSmallRound) {
            const int nr = qb - q0 < kSmallRound ? qb - q0 : kSmallRound;
            if (q0 > qa) {
                __syncthreads();  // every wave is done with the previous round's coefficients
                stage(q0, nr);
                __syncthreads();
            }
            const int ja = wave * kSmallPerWave < nr ? wave * kSmallPerWave : nr;
            const int jb = ja + kSmallPerWave < nr ? ja + kSmallPerWave : nr;
            const int n_h = (q0 + jb < q_head ? jb : (q_head - q0 > ja ? q_head - q0 : ja)) - ja;  // head queries of [ja, jb)
            unsigned long long* c0 = cnt + (q0 - qa) + ja;
            score_side<MODEL, HEAD, D>(e, valid, coef + ja * CS, kt + ja, c0, n_h, lane);
            score_side<MODEL, TAIL, D>(e, valid, coef + (ja + n_h) * CS, kt + ja + n_h, c0 + n_h, jb - ja - n_h, lane);
        }
        BLP_ST(2);
    };
    if constexpr (MULTI) {
        for (int t = slot; t < n_tiles; t += n_slots) one_tile(t, t == slot);
    } else {
        one_tile(slot, true);
    }
    // 5. the slot's partial counts (every entry is written: nothing to zero beforehand)
    __syncthreads();
    for (int j = tid; j < qb - qa; j += kWaves * 64) partial[(size_t)slot * Q + qa + j] = cnt[j];
#ifdef BLP_TIMING
    BLP_ST(3);
    if (lane == 0) {
        for (int i = 0; i < 4; ++i) atomicAdd(&g_small_timing[i], tacc[i]);
        atomicAdd(&g_small_timing[7], 1ull);
    }
#endif
}

constexpr int kSmallLdsWaves = 8;  // waves per workgroup of the TransE kernel below

// Partial sums of n <= 4 queries of one side over the lane's row (rd: its LDS address), 32 columns at a time, coefficient
// rows crow, crow + C, ... as SGPR operands.  The scalar loads are issued by hand one UNIT -- 16 columns of one query --
// ahead of their use, into a ring of two units, and drained after the arithmetic of the unit before: the compiler's own
// scalar loads are waited for right where they are issued, one exposed scalar-cache round trip per 32-48 instructions.
// (The sums start at 0 instead of at the first |difference|: 0 + |d| == |d| bit for bit, |d| being +0, positive or NaN.  The
// piece loop is not unrolled: unrolled, the compiler hoists every piece's LDS reads -- 128 registers and spills.)
template <int SIDE, int D>
__device__ __forceinline__ void sgpr_group_sums(const float* rd, const float* __restrict__ crow, int n, float (&sum)[4]) {
    constexpr int C = SIDE == TAIL ? D : 2 * D, NP = D / kSubCols;
    // Always four queries: a short group repeats its last query (results ignored by the caller) -- straight-line code, every
    // ring slot assigned exactly once per unit (conditional loads cost hundreds of spilled scalar registers).
    const float* row[4];
    static_for<4>([&](auto jj) {
        constexpr int j = decltype(jj)::value;
        row[j] = crow + (size_t)(j < n ? j : n - 1) * C;
        sum[j] = 0.f;
    });
    sf16 ra[2], rb[2];  // ring slot = half of the piece; ra: h + r (tail) or r (head); rb: t (head only)
    ra[0] = sload16<0>(row[0]);
    if constexpr (SIDE == HEAD) rb[0] = sload16<D * 4>(row[0]);
#pragma unroll 1
    for (int s = 0; s < NP; ++s) {
        float x[kSubCols];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float4 v = *reinterpret_cast<const float4*>(rd + kSubCols * s + 4 * j);
            x[4 * j] = v.x; x[4 * j + 1] = v.y; x[4 * j + 2] = v.z; x[4 * j + 3] = v.w;
        }
        // (the compiler's waits for these LDS reads belong HERE: placed at the first use, after the hand-issued scalar load
        //  of the first unit, they would drain that request as well -- LDS and scalar loads share a counter)
        asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]), "+v"(x[8]),
                     "+v"(x[9]), "+v"(x[10]), "+v"(x[11]), "+v"(x[12]), "+v"(x[13]), "+v"(x[14]), "+v"(x[15]));
        asm volatile("" : "+v"(x[16]), "+v"(x[17]), "+v"(x[18]), "+v"(x[19]), "+v"(x[20]), "+v"(x[21]), "+v"(x[22]), "+v"(x[23]),
                     "+v"(x[24]), "+v"(x[25]), "+v"(x[26]), "+v"(x[27]), "+v"(x[28]), "+v"(x[29]), "+v"(x[30]), "+v"(x[31]));
        // after the last piece: the group's first unit again (a load nobody waits for)
        const float* next_first = s + 1 < NP ? row[0] + kSubCols * (s + 1) : row[0];
        static_for<4>([&](auto jj) {
            constexpr int j = decltype(jj)::value;
            const float* cs = row[j] + kSubCols * s;
            static_for<2>([&](auto hh) {
                constexpr int h = decltype(hh)::value;
                // this unit's coefficients have landed (requested one unit ago; the wait takes the previous unit's sum as an
                // operand so that its arithmetic cannot sink below it) ...
                float& prev = sum[h == 1 ? j : (j + 3) % 4];
                if constexpr (SIDE == HEAD) asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(ra[h]), "+s"(rb[h]), "+v"(prev) : : "memory");
                else asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(ra[h]), "+v"(prev) : : "memory");
                // ... and the next unit's are requested before its arithmetic: the other half of this query's piece, the next
                // query's first half, the next piece
                if constexpr (h == 0) {
                    ra[1] = sload16<16 * 4>(cs);
                    if constexpr (SIDE == HEAD) rb[1] = sload16<(D + 16) * 4>(cs);
                } else {
                    const float* nx;
                    if constexpr (j + 1 < 4) nx = row[j + 1] + kSubCols * s; else nx = next_first;
                    ra[0] = sload16<0>(nx);
                    if constexpr (SIDE == HEAD) rb[0] = sload16<D * 4>(nx);
                }
                // (nothing of this unit's arithmetic before the requests above: left alone, the compiler schedules it ahead
                //  of them and the request is drained a few instructions after it was made)
                asm volatile("" : "+v"(sum[j]), "+v"(x[16 * h]), "+v"(x[16 * h + 1]), "+v"(x[16 * h + 2]), "+v"(x[16 * h + 3]),
                             "+v"(x[16 * h + 4]), "+v"(x[16 * h + 5]), "+v"(x[16 * h + 6]), "+v"(x[16 * h + 7]), "+v"(x[16 * h + 8]),
                             "+v"(x[16 * h + 9]), "+v"(x[16 * h + 10]), "+v"(x[16 * h + 11]), "+v"(x[16 * h + 12]),
                             "+v"(x[16 * h + 13]), "+v"(x[16 * h + 14]), "+v"(x[16 * h + 15]));
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    float d;
                    if constexpr (SIDE == TAIL) {
                        d = ra[h][k] - x[16 * h + k];              // (h + r) - e, h + r hoisted
                    } else {
                        const float y = x[16 * h + k] + ra[h][k];  // (e + r) - t
                        d = y - rb[h][k];
                    }
                    sum[j] = sum[j] + fabsf(d);
                }
            });
        });
    }
}

// ---- TransE, the tile in LDS, the coefficients in scalar registers ---------------------------------------------------------
// A lane that keeps its 128-float row in registers leaves room for two waves per SIMD, and a wave issues one VALU
// instruction per ~4.5 cycles whatever it does.  TransE consumes the row strictly left to right, so the row can stay in LDS.
// (An earlier kernel read the coefficients from LDS as well, as broadcasts next to the row quads: bound by LDS traffic,
// 28.6 us for 128 queries.)  Here the coefficients do not go through LDS at all: the true-key launch materialises the
// coefficient rows (true_key_lane_kernel's extra workgroups -- still three launches), a wave reads them through the scalar
// cache as SGPR operands (the way rank_stream.hip does), and the tile row is read 32 columns at a time -- eight
// ds_read_b128 per piece and group of FOUR queries, whose partial sums are all that crosses a piece (TransE sums left to
// right).  LDS traffic per (wave, query) drops from 2-3 KB to 128 B per lane; what is left is the VALU and the workgroup's
// start (tile -> LDS).  Eight waves per workgroup, four per SIMD; groups of four queries go round the waves; counts
// straight to the slot's partial counts.  Same operations in the same order as Scorer<TRANSE, SIDE, D>::score<false>.
template <int D>
__global__ __launch_bounds__(kSmallLdsWaves * 64, 4) void rank_small_sgpr_kernel(
    const float* __restrict__ table, int64_t N, int64_t ld, const float* __restrict__ coef_head,
    const float* __restrict__ coef_tail, const float* __restrict__ key_true, int q_head, int q_tail, int n_tiles, int n_slots,
    int q_chunk, unsigned long long* __restrict__ partial) {
    constexpr int TS = D + 4, NT = kSmallLdsWaves * 64;
    extern __shared__ __attribute__((aligned(16))) float smem[];  // the tile: 64 x TS
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    // workgroup -> (query chunk, tile).  Tables of more tiles than slots (WN18RR: 640): tiles t, t + n_slots, ... ADD to one
    // slot's counts (zeroed by the true-key launch; two or three workgroups per address: no contention to speak of)
    const int chunk = blockIdx.x / n_tiles, tile_i = blockIdx.x % n_tiles, slot = tile_i % n_slots;
    const bool shared_slot = n_tiles > n_slots;
    const int Q = q_head + q_tail;
    const int qa = chunk * q_chunk;
    const int qb = qa + q_chunk < Q ? qa + q_chunk : Q;
    const int64_t row0 = (int64_t)tile_i * kTileRows;
    {   // the tile, 16-byte pieces tid, tid + 512, ...: whole 512-byte rows per 32 threads; rows past the end: the last row
        typedef float floatx4 __attribute__((ext_vector_type(4)));
        constexpr int kPieces = kTileRows * (D / 4) / NT;
        floatx4 piece[kPieces];
        static_for<kPieces>([&](auto kk) {
            constexpr int k = decltype(kk)::value;
            const int idx = tid + k * NT, r = idx / (D / 4), c = idx % (D / 4);
            int64_t row = row0 + r;
            row = row < N ? row : N - 1;
            piece[k] = *reinterpret_cast<const floatx4*>(table + row * ld + 4 * c);
        });
        static_for<kPieces>([&](auto kk) {
            constexpr int k = decltype(kk)::value;
            const int idx = tid + k * NT, r = idx / (D / 4), c = idx % (D / 4);
            *reinterpret_cast<floatx4*>(smem + r * TS + 4 * c) = piece[k];
        });
    }
    __syncthreads();
    const bool valid = row0 + lane < N;
    const float* rd = smem + lane * TS;
    // groups of up to four queries of ONE side (the chunk's head-replacing queries, then its tail-replacing ones), group g
    // to wave g % 8
    const int nh = q_head > qa ? (q_head < qb ? q_head - qa : qb - qa) : 0;  // head-replacing queries of the chunk
    const int gh = (nh + 3) >> 2, gt = (qb - qa - nh + 3) >> 2;
    for (int g = wave; g < gh + gt; g += kSmallLdsWaves) {
        const bool head = g < gh;
        const int q0 = head ? qa + 4 * g : qa + nh + 4 * (g - gh);
        const int end = head ? qa + nh : qb;
        const int n = end - q0 < 4 ? end - q0 : 4;
        float sum[4];
        if (head) sgpr_group_sums<HEAD, D>(rd, coef_head + (size_t)q0 * 2 * D, n, sum);
        else sgpr_group_sums<TAIL, D>(rd, coef_tail + (size_t)(q0 - q_head) * D, n, sum);
        static_for<4>([&](auto jj) {
            constexpr int j = decltype(jj)::value;
            if (j < n) {
                const float key = -sum[j], kt = key_true[q0 + j];
                const unsigned long long gt_ = __popcll(__ballot(valid && key > kt)), ge_ = __popcll(__ballot(valid && key >= kt));
                if (lane == 0) {
                    if (shared_slot) atomicAdd(partial + (size_t)slot * Q + q0 + j, gt_ | (ge_ << 32));
                    else partial[(size_t)slot * Q + q0 + j] = gt_ | (ge_ << 32);
                }
            }
        });
    }
}

// Blocks this kernel takes by itself (knob small_kernel: 1 = every block it can, 2 = none).  Above a few hundred
// queries the pre-pass paths (TransE: >= kSadMinPairs pairs, bilinear: the MFMA GEMM) are faster.
bool rank_small_applicable(int model, int D, int64_t N, int64_t q_head, int64_t q_tail) {
    const long long forced = knob(KNOB_SMALL_KERNEL);
    if (forced == 2 || (D != 64 && D != 128) || knob(KNOB_RANK_KERNEL) == 1) return false;
    // (forced == 3: as 1, with TransE on the register-tile kernel instead of the scalar-register one)
    const int64_t Q = q_head + q_tail;
    if (Q == 0 || Q > kSmallMaxQueries || N <= 0) return false;
    if (forced == 1 || forced == 3) return true;
    // <= 4 + 4 queries (the reference's Wikidata5M batch) against more than one tile per slot: the streaming kernels
    // (rank_stream.hip); against a small table -- the reference's own Wikidata5M protocol ranks against the ~7.4 k entities of
    // the evaluated triples (train.py:312-314) -- this kernel (tools/few_queries_probe.py: 8 queries x 7 400 rows 31 -> 21 us)
    if (q_head <= 4 && q_tail <= 4 && N > (int64_t)kSmallMaxSlots * kTileRows) return false;
    if (model == TRANSE && knob(KNOB_SAD_MIN_QUERIES) > 0 && Q >= knob(KNOB_SAD_MIN_QUERIES)) return false;  // A/B knob
    if (model == TRANSE)  // (the scalar-register kernel -- tables of up to kSmallMaxSlots tiles -- holds out longer)
        return Q * N < (rank_small_wants_coef(model, D, N) ? kSmallMaxPairsTransESgpr : kSmallMaxPairsTransE);
    return Q * N < kSmallMaxPairsBilinear;
}

// TransE against a table of at most kSmallSgprMaxTiles tiles: rank_small_sgpr_kernel, which reads materialised coefficient
// rows (the caller has the true-key launch write them -- and zero the slots' counts when tiles share slots).  Knob small_kernel = 3: the register-tile kernel
// instead (A/B runs, tests).
bool rank_small_wants_coef(int model, int D, int64_t N) {
    const long long forced = knob(KNOB_SMALL_KERNEL);
    return model == TRANSE && (N + kTileRows - 1) / kTileRows <= kSmallSgprMaxTiles && forced != 3;
}

int rank_small_slots(int64_t N) {
    const int64_t n_tiles = (N + kTileRows - 1) / kTileRows;
    return (int)(n_tiles < kSmallMaxSlots ? n_tiles : kSmallMaxSlots);
}

template <int MODEL, int D>
static hipError_t rank_small_impl(const float* table, int64_t N, int64_t ld, const QRows q_fixed, const QRows q_rel,
                                  const float* coef_head, const float* coef_tail, const float* key_true, int64_t q_head,
                                  int64_t q_tail, unsigned long long* partial, int n_cu, hipStream_t stream) {
    const int64_t Q = q_head + q_tail;
    const int64_t n_tiles = (N + kTileRows - 1) / kTileRows;
    const int64_t n_slots = rank_small_slots(N);
    // About two workgroups per CU -- the resident set, two waves per SIMD -- and no more: (tile slot, query chunk)
    // workgroups, the chunk a whole number of rounds.
    int64_t n_chunks = (2 * (int64_t)n_cu + n_slots / 2) / n_slots;
    n_chunks = n_chunks < 1 ? 1 : n_chunks;
    int64_t q_chunk = (Q + n_chunks - 1) / n_chunks;
    q_chunk = (q_chunk + kSmallRound - 1) / kSmallRound * kSmallRound;
    if (const long long forced = knob(KNOB_EXACT_QUERY_CHUNK); forced >= 1) q_chunk = forced;
    q_chunk = q_chunk > kSmallMaxChunk ? kSmallMaxChunk : q_chunk;
    n_chunks = (Q + q_chunk - 1) / q_chunk;
    const int64_t blocks = n_slots * n_chunks;
    const size_t lds = ((size_t)kTileRows * (D + 4) + (size_t)kSmallRound * small_coef_stride(D) + kSmallRound) * 4 +
                       (size_t)kSmallMaxChunk * 8;
    if constexpr (MODEL == TRANSE) {
        if (rank_small_wants_coef(MODEL, D, N)) {  // the tile in LDS, coefficients through the scalar cache
            // 45 registers: four workgroups per CU (LDS), eight waves per SIMD -- what hides the scalar-cache latency.  One
            // round per workgroup: chunks of 32 queries ([measured] 128 queries: 33.5 us per call at 64, 30.0 at 32)
            int64_t qc = kSmallRound;
            if (const long long forced = knob(KNOB_EXACT_QUERY_CHUNK); forced >= 1) qc = forced > kSmallMaxChunk ? kSmallMaxChunk : forced;
            const int64_t grid = n_tiles * ((Q + qc - 1) / qc);
            rank_small_sgpr_kernel<D><<<dim3((unsigned)grid), kSmallLdsWaves * 64, (size_t)kTileRows * (D + 4) * 4, stream>>>(
                table, N, ld, coef_head, coef_tail, key_true, (int)q_head, (int)q_tail, (int)n_tiles, (int)n_slots, (int)qc, partial);
            return hipGetLastError();
        }
    }
    if (n_tiles > n_slots)
        rank_small_kernel<MODEL, D, true><<<dim3((unsigned)blocks), kWaves * 64, lds, stream>>>(
            table, N, ld, q_fixed, q_rel, key_true, (int)q_head, (int)q_tail, (int)n_tiles, (int)n_slots, (int)q_chunk, partial);
    else
        rank_small_kernel<MODEL, D, false><<<dim3((unsigned)blocks), kWaves * 64, lds, stream>>>(
            table, N, ld, q_fixed, q_rel, key_true, (int)q_head, (int)q_tail, (int)n_tiles, (int)n_slots, (int)q_chunk, partial);
    return hipGetLastError();
}

hipError_t launch_rank_small(int model, int D, const float* table, int64_t N, int64_t ld, const QRows q_fixed,
                             const QRows q_rel, const float* coef_head, const float* coef_tail, const float* key_true,
                             int64_t q_head, int64_t q_tail, unsigned long long* partial, int n_cu, hipStream_t stream) {
#define BLP_SMALL_CASE(M, DD) \
    case M * 1000 + DD: return rank_small_impl<M, DD>(table, N, ld, q_fixed, q_rel, coef_head, coef_tail, key_true, q_head, q_tail, partial, n_cu, stream);
    switch (model * 1000 + D) {
        BLP_SMALL_CASE(TRANSE, 64) BLP_SMALL_CASE(TRANSE, 128)
        BLP_SMALL_CASE(DISTMULT, 64) BLP_SMALL_CASE(DISTMULT, 128)
        BLP_SMALL_CASE(COMPLEX, 64) BLP_SMALL_CASE(COMPLEX, 128)
        BLP_SMALL_CASE(SIMPLE, 64) BLP_SMALL_CASE(SIMPLE, 128)
    default: return hipErrorInvalidValue;
    }
#undef BLP_SMALL_CASE
}

}  // namespace blp

#ifdef BLP_TIMING
extern "C" int blp_debug_read_small_timing(unsigned long long* out) {
    hipError_t err = hipMemcpyFromSymbol(out, HIP_SYMBOL(blp::g_small_timing), sizeof(blp::g_small_timing));
    unsigned long long zero[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (err == hipSuccess) err = hipMemcpyToSymbol(HIP_SYMBOL(blp::g_small_timing), zero, sizeof(zero));
    return (int)err;
}
#endif
