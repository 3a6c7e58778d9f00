// valu_ubench.hip -- what f32 VALU issue rate does gfx950 sustain for the instruction mixes of the
// TransE rank kernel (dependent L1 chain, VOP3 |x| modifier, DPP quad_perm operand), at 1..4 waves
// per SIMD?  Reports shader cycles per wave-instruction per SIMD (peak would be 2.0 on a SIMD-32).
// Build+run on the GPU box: hipcc --offload-arch=gfx950 -O3 -o /tmp/valu tools/valu_ubench.hip && /tmp/valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>

#define CK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(err_), __LINE__); return 1; } } while (0)
#define R4(x) x x x x
#define R16(x) R4(R4(x))

// each body = 64 VALU instructions
#define BODY_INDEP4 R16("v_add_f32 v0, v0, v8\n v_add_f32 v1, v1, v8\n v_add_f32 v2, v2, v8\n v_add_f32 v3, v3, v8\n")
#define BODY_DEP1 R16(R4("v_add_f32 v0, v0, v8\n"))
#define BODY_INDEP4_ABS R16("v_add_f32_e64 v0, v0, |v8|\n v_add_f32_e64 v1, v1, |v8|\n v_add_f32_e64 v2, v2, |v8|\n v_add_f32_e64 v3, v3, |v8|\n")
#define BODY_PAIR_NAIVE R16(R4("v_sub_f32 v4, v9, v10\n v_add_f32_e64 v0, v0, |v4|\n"))
#define BODY_PAIR_NAIVE_H R4(R4(R4("v_sub_f32 v4, v9, v10\n v_add_f32_e64 v0, v0, |v4|\n")))
#define BODY_PAIR_PIPE R4(R4("v_sub_f32 v4, v9, v10\n v_add_f32_e64 v0, v0, |v5|\n v_sub_f32 v5, v9, v11\n v_add_f32_e64 v0, v0, |v6|\n v_sub_f32 v6, v9, v12\n v_add_f32_e64 v0, v0, |v7|\n v_sub_f32 v7, v9, v13\n v_add_f32_e64 v0, v0, |v4|\n"))
#define DPP " quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
#define BODY_PAIR_PIPE_DPP R4(R4("v_sub_f32_dpp v4, v9, v10" DPP " v_add_f32_e64 v0, v0, |v5|\n v_sub_f32_dpp v5, v9, v11" DPP " v_add_f32_e64 v0, v0, |v6|\n v_sub_f32_dpp v6, v9, v12" DPP " v_add_f32_e64 v0, v0, |v7|\n v_sub_f32_dpp v7, v9, v13" DPP " v_add_f32_e64 v0, v0, |v4|\n"))
#define BODY_2CHAIN_PIPE_DPP R4(R4("v_sub_f32_dpp v4, v9, v10" DPP " v_add_f32_e64 v0, v0, |v5|\n v_sub_f32_dpp v5, v14, v10" DPP " v_add_f32_e64 v1, v1, |v6|\n v_sub_f32_dpp v6, v9, v12" DPP " v_add_f32_e64 v0, v0, |v7|\n v_sub_f32_dpp v7, v14, v12" DPP " v_add_f32_e64 v1, v1, |v4|\n"))
#define BODY_FMA_INDEP4 R16("v_fma_f32 v0, v0, v8, v9\n v_fma_f32 v1, v1, v8, v9\n v_fma_f32 v2, v2, v8, v9\n v_fma_f32 v3, v3, v8, v9\n")
#define BODY_PK_ADD R16("v_pk_add_f32 v[0:1], v[0:1], v[8:9]\n v_pk_add_f32 v[2:3], v[2:3], v[8:9]\n v_pk_add_f32 v[4:5], v[4:5], v[8:9]\n v_pk_add_f32 v[6:7], v[6:7], v[8:9]\n")

#define KERNEL(name, body)                                                                          \
    __global__ __launch_bounds__(64) void name(int iters, unsigned long long* out) {                \
        unsigned long long t0 = __builtin_readcyclecounter();                                       \
        for (int i = 0; i < iters; ++i)                                                             \
            asm volatile(body ::: "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", \
                         "v11", "v12", "v13", "v14", "v15");                                        \
        unsigned long long t1 = __builtin_readcyclecounter();                                       \
        if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;                                            \
    }

KERNEL(k_indep4, BODY_INDEP4)
KERNEL(k_dep1, BODY_DEP1)
KERNEL(k_indep4_abs, BODY_INDEP4_ABS)
KERNEL(k_pair_naive, BODY_PAIR_NAIVE)
KERNEL(k_pair_pipe, BODY_PAIR_PIPE)
KERNEL(k_pair_pipe_dpp, BODY_PAIR_PIPE_DPP)
KERNEL(k_2chain_pipe_dpp, BODY_2CHAIN_PIPE_DPP)
KERNEL(k_fma_indep4, BODY_FMA_INDEP4)
KERNEL(k_pk_add, BODY_PK_ADD)

typedef void (*kern_t)(int, unsigned long long*);

int run(const char* name, kern_t k, int waves_per_simd, int insts_per_body) {
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
    static unsigned long long h[4096 * 4];
    CK(hipMemcpy(h, d, blocks * 8, hipMemcpyDeviceToHost));
    double mean = 0; for (int i = 0; i < blocks; ++i) mean += (double)h[i]; mean /= blocks;
    const double insts = (double)iters * insts_per_body;
    // readcyclecounter ticks at a fixed 100 MHz on gfx9 (s_memtime = shader clock on gfx950 per guide);
    // report wall-clock based rate too
    const double total = insts * blocks;  // wave-instructions
    printf("%-22s waves/SIMD=%d  %.3f ms  %.1f G wave-inst/s  -> %.2f cyc/inst/SIMD @2.4GHz   (counter/inst per wave %.2f)\n",
           name, waves_per_simd, ms, total / ms / 1e6, 2.4e9 / (total / (ms * 1e-3) / 1024), mean / insts);
    CK(hipFree(d));
    return 0;
}

int main() {
    for (int w = 1; w <= 4; ++w) {
        run("indep4 add", k_indep4, w, 64);
        run("dep1 add", k_dep1, w, 64);
        run("indep4 add |abs|", k_indep4_abs, w, 64);
        run("pair naive sub;add|x|", k_pair_naive, w, 128);
        run("pair pipelined", k_pair_pipe, w, 128);
        run("pair pipelined dpp", k_pair_pipe_dpp, w, 128);
        run("2chain pipelined dpp", k_2chain_pipe_dpp, w, 128);
        run("indep4 fma", k_fma_indep4, w, 64);
        run("pk_add (2 flop/lane)", k_pk_add, w, 64);
        printf("\n");
    }
    return 0;
}
