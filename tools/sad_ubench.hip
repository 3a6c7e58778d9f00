// sad_ubench.hip -- issue rate of the integer sum-of-absolute-differences instructions on gfx950
// (v_sad_u16 = two |a-b| terms + accumulate per instruction), with VGPR and SGPR operands, next to
// f32 add as the yardstick.  Decides whether a quantised TransE pre-pass can beat the f32 VALU roof.
// Build+run on the GPU box: hipcc --offload-arch=gfx950 -O3 -o /tmp/sad tools/sad_ubench.hip && /tmp/sad
#include <hip/hip_runtime.h>
#include <cstdio>

#define CK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(err_), __LINE__); return 1; } } while (0)
#define R4(x) x x x x
#define R16(x) R4(R4(x))

#define BODY_ADD R16("v_add_f32 v0, v0, v8\n v_add_f32 v1, v1, v8\n v_add_f32 v2, v2, v8\n v_add_f32 v3, v3, v8\n")
#define BODY_SAD16 R16("v_sad_u16 v0, v8, v9, v0\n v_sad_u16 v1, v8, v10, v1\n v_sad_u16 v2, v8, v11, v2\n v_sad_u16 v3, v8, v12, v3\n")
#define BODY_SAD16_DEP R16(R4("v_sad_u16 v0, v8, v9, v0\n"))
#define BODY_SAD16_S R16("v_sad_u16 v0, s4, v9, v0\n v_sad_u16 v1, s5, v10, v1\n v_sad_u16 v2, s6, v11, v2\n v_sad_u16 v3, s7, v12, v3\n")
#define BODY_SAD16_S_DEP R16("v_sad_u16 v0, s4, v9, v0\n v_sad_u16 v0, s5, v10, v0\n v_sad_u16 v0, s6, v11, v0\n v_sad_u16 v0, s7, v12, v0\n")
#define BODY_SAD8 R16("v_sad_u8 v0, v8, v9, v0\n v_sad_u8 v1, v8, v10, v1\n v_sad_u8 v2, v8, v11, v2\n v_sad_u8 v3, v8, v12, v3\n")
#define BODY_SAD32 R16("v_sad_u32 v0, v8, v9, v0\n v_sad_u32 v1, v8, v10, v1\n v_sad_u32 v2, v8, v11, v2\n v_sad_u32 v3, v8, v12, v3\n")
#define BODY_PKSUB16 R16("v_pk_sub_u16 v0, v8, v9\n v_pk_sub_u16 v1, v8, v10\n v_pk_sub_u16 v2, v8, v11\n v_pk_sub_u16 v3, v8, v12\n")
#define BODY_PKADDF16 R16("v_pk_add_f16 v0, v0, v9\n v_pk_add_f16 v1, v1, v10\n v_pk_add_f16 v2, v2, v11\n v_pk_add_f16 v3, v3, v12\n")
#define BODY_DOT4 R16("v_dot4_i32_i8 v0, v8, v9, v0\n v_dot4_i32_i8 v1, v8, v10, v1\n v_dot4_i32_i8 v2, v8, v11, v2\n v_dot4_i32_i8 v3, v8, v12, v3\n")
#define BODY_CMP R16("v_cmp_lt_u32 vcc, v0, v8\n v_addc_co_u32 v1, vcc, 0, v1, vcc\n v_cmp_lt_u32 vcc, v2, v8\n v_addc_co_u32 v3, vcc, 0, v3, vcc\n")

#define KERNEL(name, body)                                                                          \
    __global__ __launch_bounds__(64) void name(int iters, unsigned long long* out) {                \
        unsigned long long t0 = __builtin_readcyclecounter();                                       \
        for (int i = 0; i < iters; ++i)                                                             \
            asm volatile(body ::: "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", \
                         "v11", "v12", "v13", "v14", "v15", "s4", "s5", "s6", "s7", "vcc");         \
        unsigned long long t1 = __builtin_readcyclecounter();                                       \
        if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;                                            \
    }

KERNEL(k_add, BODY_ADD)
KERNEL(k_sad16, BODY_SAD16)
KERNEL(k_sad16_dep, BODY_SAD16_DEP)
KERNEL(k_sad16_s, BODY_SAD16_S)
KERNEL(k_sad16_s_dep, BODY_SAD16_S_DEP)
KERNEL(k_sad8, BODY_SAD8)
KERNEL(k_sad32, BODY_SAD32)
KERNEL(k_pksub16, BODY_PKSUB16)
KERNEL(k_pkaddf16, BODY_PKADDF16)
KERNEL(k_dot4, BODY_DOT4)
KERNEL(k_cmp, BODY_CMP)

typedef void (*kern_t)(int, unsigned long long*);

int run(const char* name, kern_t k, int waves_per_simd) {
    const int blocks = 256 * 4 * waves_per_simd, iters = 2000;
    unsigned long long* d;
    CK(hipMalloc(&d, blocks * 8));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    k<<<blocks, 64>>>(10, d);
    CK(hipEventRecord(a));
    k<<<blocks, 64>>>(iters, d);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    const double total = (double)iters * 64 * blocks;  // wave-instructions
    printf("%-26s waves/SIMD=%d  %.3f ms  -> %.2f cyc/inst/SIMD @2.4GHz\n", name, waves_per_simd, ms,
           2.4e9 / (total / (ms * 1e-3) / 1024));
    CK(hipFree(d));
    return 0;
}

int main() {
    for (int w = 1; w <= 4; w *= 2) {
        run("add f32 indep4", k_add, w);
        run("sad_u16 indep4", k_sad16, w);
        run("sad_u16 dep1", k_sad16_dep, w);
        run("sad_u16 sgpr indep4", k_sad16_s, w);
        run("sad_u16 sgpr dep1", k_sad16_s_dep, w);
        run("sad_u8 indep4", k_sad8, w);
        run("sad_u32 indep4", k_sad32, w);
        run("pk_sub_u16 indep4", k_pksub16, w);
        run("pk_add_f16 indep4", k_pkaddf16, w);
        run("dot4_i32_i8 indep4", k_dot4, w);
        run("cmp+addc x2", k_cmp, w);
    }
    return 0;
}
