// mfma_exact_check.hip -- is v_mfma_f32_32x32x2_f32 with A = (e, 1), B = (-1, c) bit-identical to
// RN(c - e) (and B = (+1, r) to RN(e + r), B = (c, 0) to RN(e * c)) for arbitrary f32 inputs, incl.
// subnormals, signed zeros, huge/tiny magnitudes?  Prints the mismatch counts.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(err_), __LINE__); return 1; } } while (0)

// mode 0: c - e   mode 1: e + c   mode 2: e * c
__global__ void k(const float* e, const float* c, float* out, int mode) {
    const int lane = threadIdx.x;
    const float a = lane < 32 ? e[blockIdx.x * 32 + lane] : (mode == 2 ? 0.0f : 1.0f);
    float b;
    if (mode == 0) b = lane < 32 ? -1.0f : c[blockIdx.x * 32 + lane - 32];
    else if (mode == 1) b = lane < 32 ? 1.0f : c[blockIdx.x * 32 + lane - 32];
    else b = lane < 32 ? c[blockIdx.x * 32 + lane] : 0.0f;
    f32x16 z = {0};
    f32x16 d = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, z, 0, 0, 0);
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), col = lane & 31;
        out[(blockIdx.x * 32 + row) * 32 + col] = d[r];
    }
}
static float rnd(int kind) {
    union { unsigned u; float f; } v;
    switch (kind) {
    case 0: return (float)rand() / RAND_MAX * 2 - 1;
    case 1: v.u = ((unsigned)rand() << 16) ^ (unsigned)rand(); if (((v.u >> 23) & 0xff) == 0xff) v.u &= 0x7f7fffff; return v.f;   // any finite bits
    case 2: v.u = (rand() & 0x7fffff) | ((rand() & 1u) << 31); return v.f;   // subnormal
    default: return (rand() & 1) ? 0.0f : -0.0f;
    }
}
int main() {
    const int T = 512;  // tiles
    std::vector<float> e(T * 32), c(T * 32), out(T * 1024);
    srand(1);
    for (int i = 0; i < T * 32; ++i) { e[i] = rnd((i / 32) % 4 == 3 ? rand() % 4 : (i / 32) % 4); c[i] = rnd((i / 64) % 4 == 3 ? rand() % 4 : (i / 64) % 4); }
    float *de, *dc, *dout;
    CK(hipMalloc(&de, T * 128)); CK(hipMalloc(&dc, T * 128)); CK(hipMalloc(&dout, T * 4096));
    CK(hipMemcpy(de, e.data(), T * 128, hipMemcpyHostToDevice)); CK(hipMemcpy(dc, c.data(), T * 128, hipMemcpyHostToDevice));
    for (int mode = 0; mode < 3; ++mode) {
        k<<<T, 64>>>(de, dc, dout, mode);
        CK(hipMemcpy(out.data(), dout, T * 4096, hipMemcpyDeviceToHost));
        long bad = 0, bad_abs = 0, total = 0;
        for (int t = 0; t < T; ++t) for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
            volatile float ei = e[t * 32 + i], cj = c[t * 32 + j];
            volatile float want = mode == 0 ? cj - ei : mode == 1 ? ei + cj : ei * cj;
            float got = out[(t * 32 + i) * 32 + j];
            float w = want;
            ++total;
            if (memcmp(&got, &w, 4) != 0 && !(std::isnan(got) && std::isnan(w))) {
                ++bad;
                float ga = fabsf(got), wa = fabsf(w);
                if (memcmp(&ga, &wa, 4) != 0) { if (bad_abs < 5) printf("  mode %d: e=%a c=%a got=%a want=%a\n", mode, ei, cj, got, w); ++bad_abs; }
            }
        }
        printf("mode %d: %ld / %ld bitwise mismatches, %ld differ in |value| (the rest are sign-of-zero)\n", mode, bad, total, bad_abs);
    }
    return 0;
}
