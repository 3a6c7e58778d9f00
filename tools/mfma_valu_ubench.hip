// mfma_valu_ubench.hip -- how do v_mfma_f32_32x32x2_f32 (16-pass) and plain VALU adds overlap on one
// SIMD?  Per step: 1 MFMA (double-buffered results) + NV VALU adds reading the previous result.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(err_), __LINE__); return 1; } } while (0)
#define A16(src) \
  "v_add_f32_e64 v40, v40, |v" #src "|\n" "v_add_f32_e64 v41, v41, |v" #src "|\n" "v_add_f32_e64 v42, v42, |v" #src "|\n" "v_add_f32_e64 v43, v43, |v" #src "|\n" \
  "v_add_f32_e64 v44, v44, |v" #src "|\n" "v_add_f32_e64 v45, v45, |v" #src "|\n" "v_add_f32_e64 v46, v46, |v" #src "|\n" "v_add_f32_e64 v47, v47, |v" #src "|\n" \
  "v_add_f32_e64 v48, v48, |v" #src "|\n" "v_add_f32_e64 v49, v49, |v" #src "|\n" "v_add_f32_e64 v50, v50, |v" #src "|\n" "v_add_f32_e64 v51, v51, |v" #src "|\n" \
  "v_add_f32_e64 v52, v52, |v" #src "|\n" "v_add_f32_e64 v53, v53, |v" #src "|\n" "v_add_f32_e64 v54, v54, |v" #src "|\n" "v_add_f32_e64 v55, v55, |v" #src "|\n"
// two steps: MFMA->v[0:15] ; adds on v16.. ; MFMA->v[16:31] ; adds on v0..
#define STEP2_16 "v_mfma_f32_32x32x2_f32 v[0:15], v32, v33, 0\n" "s_nop 4\n" A16(16) "v_mfma_f32_32x32x2_f32 v[16:31], v32, v33, 0\n" "s_nop 4\n" A16(0)
#define STEP2_32 "v_mfma_f32_32x32x2_f32 v[0:15], v32, v33, 0\n" A16(16) A16(17) "v_mfma_f32_32x32x2_f32 v[16:31], v32, v33, 0\n" A16(0) A16(1)
#define STEP2_0 "v_mfma_f32_32x32x2_f32 v[0:15], v32, v33, 0\n" "v_mfma_f32_32x32x2_f32 v[16:31], v32, v33, 0\n"
#define STEP2_VONLY16 A16(16) A16(0)
#define CLOB "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55"
#define KERNEL(name, body) \
    __global__ __launch_bounds__(64) void name(int iters, float* out) { \
        for (int i = 0; i < iters; ++i) asm volatile(body body body body ::: CLOB); \
        if (threadIdx.x == 1234) out[0] = 1.f; }
KERNEL(k_mfma_only, STEP2_0)
KERNEL(k_mfma_16, STEP2_16)
KERNEL(k_mfma_32, STEP2_32)
KERNEL(k_valu16_only, STEP2_VONLY16)
typedef void (*kern_t)(int, float*);
int run(const char* name, kern_t k, int w) {
    const int blocks = 256 * 4 * w, iters = 500;
    float* d; CK(hipMalloc(&d, 4));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    k<<<blocks, 64>>>(10, d);
    CK(hipEventRecord(a)); k<<<blocks, 64>>>(iters, d); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    const double steps = (double)iters * 8 * w;  // steps per SIMD
    printf("%-18s waves/SIMD=%d  %.3f ms  -> %.1f cycles per step per SIMD @2.4GHz\n", name, w, ms, ms * 1e-3 * 2.4e9 / steps);
    CK(hipFree(d)); return 0;
}
int main() {
    for (int w = 1; w <= 4; ++w) {
        run("mfma only", k_mfma_only, w);
        run("mfma + 16 add", k_mfma_16, w);
        run("mfma + 32 add", k_mfma_32, w);
        run("16 add only", k_valu16_only, w);
        printf("\n");
    }
    return 0;
}
