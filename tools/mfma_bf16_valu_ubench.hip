// mfma_bf16_valu_ubench.hip -- how much VALU / LDS-read work hides behind v_mfma_f32_32x32x16_bf16 (8
// passes, 32 cycles of matrix pipe) on one SIMD?  Per step: 1 MFMA (two alternating accumulators) + NV
// independent f32 VALU instructions (sub / fma / cmp mix like the rank epilogue), at 1..4 waves/SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(err_), __LINE__); return 1; } } while (0)
#define V4 "v_sub_f32 v40, v41, v42\n v_fma_f32 v43, v44, v45, v46\n v_cmp_gt_f32 vcc, v47, v48\n v_addc_co_u32 v49, vcc, 0, v49, vcc\n"
#define MF0 "v_mfma_f32_32x32x16_bf16 v[0:15], v[32:35], v[36:39], v[0:15]\n"
#define MF1 "v_mfma_f32_32x32x16_bf16 v[16:31], v[32:35], v[36:39], v[16:31]\n"
#define STEP2_0 MF0 MF1
#define STEP2_4 MF0 V4 MF1 V4
#define STEP2_8 MF0 V4 V4 MF1 V4 V4
#define STEP2_12 MF0 V4 V4 V4 MF1 V4 V4 V4
#define STEP2_V8 V4 V4 V4 V4
#define CLOB "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39","v40","v43","v49","vcc"
#define KERNEL(name, body) \
    __global__ __launch_bounds__(64) void name(int iters, float* out) { \
        for (int i = 0; i < iters; ++i) asm volatile(body body body body ::: CLOB); \
        if (threadIdx.x == 1234) out[0] = 1.f; }
KERNEL(k_mfma_only, STEP2_0)
KERNEL(k_mfma_4, STEP2_4)
KERNEL(k_mfma_8, STEP2_8)
KERNEL(k_mfma_12, STEP2_12)
KERNEL(k_valu8_only, STEP2_V8)
typedef void (*kern_t)(int, float*);
int run(const char* name, kern_t k, int w) {
    const int blocks = 256 * 4 * w, iters = 500;
    float* d; CK(hipMalloc(&d, 4));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    k<<<blocks, 64>>>(10, d);
    CK(hipEventRecord(a)); k<<<blocks, 64>>>(iters, d); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    const double steps = (double)iters * 8 * w;  // MFMA steps per SIMD
    printf("%-22s waves/SIMD=%d  %.3f ms  -> %.1f cycles per step per SIMD @2.4GHz\n", name, w, ms, ms * 1e-3 * 2.4e9 / steps);
    CK(hipFree(d)); return 0;
}
int main() {
    for (int w = 1; w <= 4; ++w) {
        run("mfma only", k_mfma_only, w);
        run("mfma + 4 valu", k_mfma_4, w);
        run("mfma + 8 valu", k_mfma_8, w);
        run("mfma + 12 valu", k_mfma_12, w);
        run("8 valu only (per step)", k_valu8_only, w);
        printf("\n");
    }
    return 0;
}
