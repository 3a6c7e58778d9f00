// mfma_bf16_accum_check.hip -- how does v_mfma_f32_32x32x16_bf16 round?  For random bf16 A (32 x 16),
// B (16 x 32) and f32 C, D[i][n] is compared, bit for bit, with candidate models evaluated on the host in
// long double (64-bit significand: the exact value of 16 products of 16-bit significands plus an f32):
//   single  RN_f32( c + sum_k a_k b_k )                      one rounding per MFMA
//   seq     c, then += a_k b_k for k = 0..15, RN_f32 after every addition (an fma chain)
// and the error of D against the exact value is reported in units of ulp(result) and of
// u * (|c| + sum |a_k b_k|)  (u = 2^-24) -- the quantity the bilinear pre-pass band is priced in.
// Operand magnitudes span 2^-20 .. 2^20 with random signs, so cancellation is the normal case.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(err_), __LINE__); return 1; } } while (0)

__global__ void k(const unsigned short* a, const unsigned short* b, const float* c, float* d) {
    // A[i][k]: lane = i + 32 (k / 8), element k % 8; B[k][n]: lane = n + 32 (k / 8), element k % 8
    const int lane = threadIdx.x, t = blockIdx.x;
    unsigned short av[8], bv[8];
    for (int j = 0; j < 8; ++j) {
        const int kk = 8 * (lane >> 5) + j;
        av[j] = a[(t * 32 + (lane & 31)) * 16 + kk];
        bv[j] = b[(t * 16 + kk) * 32 + (lane & 31)];
    }
    bf16x8 A, B;
    memcpy(&A, av, 16); memcpy(&B, bv, 16);
    f32x16 C;
    for (int r = 0; r < 16; ++r) C[r] = c[(t * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 32 + (lane & 31)];
    const f32x16 D = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, C, 0, 0, 0);
    for (int r = 0; r < 16; ++r) d[(t * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 32 + (lane & 31)] = D[r];
}
static float bf2f(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }
static unsigned short rnd_bf16(int spread) {
    const int e = 127 + (rand() % (2 * spread + 1)) - spread;
    return (unsigned short)(((rand() & 1) << 15) | (e << 7) | (rand() & 0x7f));
}
int main() {
    const int T = 256;
    srand(7);
    for (int spread : {0, 4, 20}) {
        std::vector<unsigned short> a(T * 32 * 16), b(T * 16 * 32);
        std::vector<float> c(T * 1024), d(T * 1024);
        for (auto& x : a) x = rnd_bf16(spread);
        for (auto& x : b) x = rnd_bf16(spread);
        for (auto& x : c) x = bf2f(rnd_bf16(spread)) * (1.0f + (float)rand() / RAND_MAX * 0.01f);
        unsigned short *da, *db; float *dc, *dd;
        CK(hipMalloc(&da, a.size() * 2)); CK(hipMalloc(&db, b.size() * 2)); CK(hipMalloc(&dc, c.size() * 4)); CK(hipMalloc(&dd, d.size() * 4));
        CK(hipMemcpy(da, a.data(), a.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(db, b.data(), b.size() * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(dc, c.data(), c.size() * 4, hipMemcpyHostToDevice));
        k<<<T, 64>>>(da, db, dc, dd);
        CK(hipMemcpy(d.data(), dd, d.size() * 4, hipMemcpyDeviceToHost));
        long n = 0, eq_single = 0, eq_seq = 0;
        double worst_ulp = 0, worst_u_sum = 0;
        for (int t = 0; t < T; ++t) for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
            long double exact = c[(t * 32 + i) * 32 + j], mag = fabsl(exact);
            float seq = c[(t * 32 + i) * 32 + j];
            for (int kk = 0; kk < 16; ++kk) {
                const float x = bf2f(a[(t * 32 + i) * 16 + kk]), y = bf2f(b[(t * 16 + kk) * 32 + j]);
                exact += (long double)x * y; mag += fabsl((long double)x * y);
                seq = fmaf(x, y, seq);
            }
            const float single = (float)exact, got = d[(t * 32 + i) * 32 + j];
            ++n; eq_single += memcmp(&single, &got, 4) == 0; eq_seq += memcmp(&seq, &got, 4) == 0;
            const long double err = fabsl((long double)got - exact);
            const double ulp = ldexp(1.0, ilogb((double)fabsl(exact) + 1e-300) - 23);
            worst_ulp = fmax(worst_ulp, (double)(err / ulp));
            worst_u_sum = fmax(worst_u_sum, (double)(err / (5.9604644775390625e-8L * mag)));
        }
        printf("exponent spread +-%2d: %ld outputs; == single rounding %.2f %%, == fma chain %.2f %%; worst error %.3f ulp(result), %.4f x u x (|c| + sum|a b|)\n",
               spread, n, 100.0 * eq_single / n, 100.0 * eq_seq / n, worst_ulp, worst_u_sum);
        hipFree(da); hipFree(db); hipFree(dc); hipFree(dd);
    }
    // adversarial: c just above a power of two, sixteen equal products far below ulp(c): how much of them survives?
    for (int shift = 20; shift <= 32; ++shift) {
        std::vector<unsigned short> a(32 * 16), b(16 * 32);
        std::vector<float> c(1024), d(1024);
        const unsigned short one = 0x3f80;                                  // 1.0
        const unsigned short tiny = (unsigned short)(((127 - shift) << 7) | 0x7f);  // 1.9921875 * 2^-shift
        for (auto& x : a) x = one;
        for (auto& x : b) x = tiny;
        for (auto& x : c) x = 1.0f;
        unsigned short *da, *db; float *dc, *dd;
        CK(hipMalloc(&da, a.size() * 2)); CK(hipMalloc(&db, b.size() * 2)); CK(hipMalloc(&dc, c.size() * 4)); CK(hipMalloc(&dd, d.size() * 4));
        CK(hipMemcpy(da, a.data(), a.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(db, b.data(), b.size() * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(dc, c.data(), c.size() * 4, hipMemcpyHostToDevice));
        k<<<1, 64>>>(da, db, dc, dd);
        CK(hipMemcpy(d.data(), dd, d.size() * 4, hipMemcpyDeviceToHost));
        const long double exact = 1.0L + 16.0L * bf2f(tiny), mag = exact;
        printf("c = 1, 16 products of 1.992 * 2^-%d: D - exact = %+.3f ulp(1) = %+.3f x u x (|c| + sum|a b|)\n", shift,
               (double)(((long double)d[0] - exact) / 1.1920928955078125e-7L), (double)(((long double)d[0] - exact) / (5.9604644775390625e-8L * mag)));
        hipFree(da); hipFree(db); hipFree(dc); hipFree(dd);
    }
    return 0;
}
