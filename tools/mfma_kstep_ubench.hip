// mfma_kstep_ubench.hip -- the K-step of the bilinear pre-pass in isolation: v_mfma_f32_32x32x16_bf16 on 2 or 4
// independent accumulators, interleaved with the decision VALU of rank_gemm.hip (v_cmp -> SGPR pair, v_addc reading it:
// 4 VALU per MFMA in the main phase) and, optionally, one LDS operand read + wait per K-step.  Cycles per MFMA at 1 and
// 2 waves per SIMD (the pipe needs 32).   hipcc --offload-arch=gfx950 -O3 -o /tmp/ks tools/mfma_kstep_ubench.hip && /tmp/ks
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(err_), __LINE__); return 1; } } while (0)
#define C4 "v_cmp_gt_f32_e64 s[84:85], v100, v101\n v_cmp_lt_f32_e64 s[86:87], v100, v102\n v_cmp_gt_f32_e64 s[88:89], v103, v101\n v_cmp_lt_f32_e64 s[90:91], v103, v102\n"
#define A4 "v_addc_co_u32_e64 v104, s[84:85], v104, v104, s[84:85]\n v_addc_co_u32_e64 v105, s[86:87], v105, v105, s[86:87]\n v_addc_co_u32_e64 v104, s[88:89], v104, v104, s[88:89]\n v_addc_co_u32_e64 v105, s[90:91], v105, v105, s[90:91]\n"
#define MF(a) "v_mfma_f32_32x32x16_bf16 v[" a "], v[96:99], v[108:111], v[" a "]\n"
#define LDS "ds_read_b128 v[108:111], v112\n s_waitcnt lgkmcnt(0)\n"
#define K2 MF("0:15") C4 MF("16:31") A4
#define K4 MF("0:15") C4 MF("16:31") A4 MF("32:47") C4 MF("48:63") A4
#define K2L LDS K2
#define K4L LDS MF("0:15") C4 MF("16:31") A4 LDS MF("32:47") C4 MF("48:63") A4
#define K4M MF("0:15") MF("16:31") MF("32:47") MF("48:63")
#define CLOB "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63","v96","v97","v98","v99","v100","v101","v102","v103","v104","v105","v108","v109","v110","v111","v112","s84","s85","s86","s87","s88","s89","s90","s91"
#define KERNEL(name, body, reps) \
    __global__ __launch_bounds__(64) void name(int iters, float* out) { \
        __shared__ float lds[1024]; lds[threadIdx.x] = 1.f; \
        asm volatile("v_mov_b32 v112, 0" ::: "v112"); \
        for (int i = 0; i < iters; ++i) asm volatile(reps ::: CLOB, "memory"); \
        if (threadIdx.x == 1234) out[0] = lds[5]; }
KERNEL(k2, K2, K2 K2 K2 K2 K2 K2 K2 K2)       // 16 MFMAs per iteration
KERNEL(k4, K4, K4 K4 K4 K4)
KERNEL(k2l, K2L, K2L K2L K2L K2L K2L K2L K2L K2L)
KERNEL(k4l, K4L, K4L K4L K4L K4L)
KERNEL(k4m, K4M, K4M K4M K4M K4M)
typedef void (*kern_t)(int, float*);
int run(const char* name, kern_t k, int w) {
    const int blocks = 256 * 4 * w, iters = 400;
    float* d; CK(hipMalloc(&d, 4));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    k<<<blocks, 64>>>(10, d);
    CK(hipEventRecord(a)); k<<<blocks, 64>>>(iters, d); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    const double mfmas = (double)iters * 16 * w;  // per SIMD
    printf("%-34s waves/SIMD=%d  %.3f ms  -> %.1f cycles per MFMA per SIMD @2.4GHz\n", name, w, ms, ms * 1e-3 * 2.4e9 / mfmas);
    CK(hipFree(d)); return 0;
}
int main() {
    for (int w = 1; w <= 2; ++w) {
        run("4 accumulators, MFMA only", k4m, w);
        run("2 accumulators + 4 VALU/MFMA", k2, w);
        run("4 accumulators + 4 VALU/MFMA", k4, w);
        run("2 acc + 4 VALU/MFMA + LDS read", k2l, w);
        run("4 acc + 4 VALU/MFMA + LDS read", k4l, w);
        printf("\n");
    }
    return 0;
}
