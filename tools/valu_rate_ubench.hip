// valu_rate_ubench.hip -- issue rates of the VALU instructions the streaming kernels' arithmetic is made of, per encoding:
// plain f32 add (VOP2), the same with an |abs| source modifier (VOP3), fma, f16 -> f32 conversion, the shift / mask a bfloat16
// element widens with, basic 32-bit integer ops, and v_fma_mix_f32 (an f16 source converted inside the instruction).
// Build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 -o /tmp/vr tools/valu_rate_ubench.hip && /tmp/vr
#include <hip/hip_runtime.h>
#include <cstdio>

#define CK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(err_), __LINE__); return 1; } } while (0)
#define R4(x) x x x x
#define R16(x) R4(R4(x))

#define B_ADD R16("v_add_f32 v0, v0, v8\n v_add_f32 v1, v1, v8\n v_add_f32 v2, v2, v8\n v_add_f32 v3, v3, v8\n")
#define B_ADD_ABS R16("v_add_f32 v0, v0, |v8|\n v_add_f32 v1, v1, |v9|\n v_add_f32 v2, v2, |v10|\n v_add_f32 v3, v3, |v11|\n")
#define B_SUB_S R16("v_sub_f32 v0, s4, v8\n v_sub_f32 v1, s5, v9\n v_sub_f32 v2, s6, v10\n v_sub_f32 v3, s7, v11\n")
#define B_FMA R16("v_fma_f32 v0, v8, v9, v0\n v_fma_f32 v1, v8, v10, v1\n v_fma_f32 v2, v8, v11, v2\n v_fma_f32 v3, v8, v12, v3\n")
#define B_FMA_S R16("v_fma_f32 v0, v8, s4, v0\n v_fma_f32 v1, v9, s5, v1\n v_fma_f32 v2, v10, s6, v2\n v_fma_f32 v3, v11, s7, v3\n")
#define B_CVT R16("v_cvt_f32_f16 v0, v8\n v_cvt_f32_f16 v1, v9\n v_cvt_f32_f16 v2, v10\n v_cvt_f32_f16 v3, v11\n")
#define B_CVT_HI R16("v_cvt_f32_f16_sdwa v0, v8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n v_cvt_f32_f16_sdwa v1, v9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n v_cvt_f32_f16_sdwa v2, v10 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n v_cvt_f32_f16_sdwa v3, v11 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n")
#define B_LSHL R16("v_lshlrev_b32 v0, 16, v8\n v_lshlrev_b32 v1, 16, v9\n v_lshlrev_b32 v2, 16, v10\n v_lshlrev_b32 v3, 16, v11\n")
#define B_AND R16("v_and_b32 v0, 0xffff0000, v8\n v_and_b32 v1, 0xffff0000, v9\n v_and_b32 v2, 0xffff0000, v10\n v_and_b32 v3, 0xffff0000, v11\n")
#define B_ADDU R16("v_add_u32 v0, v0, v8\n v_add_u32 v1, v1, v8\n v_add_u32 v2, v2, v8\n v_add_u32 v3, v3, v8\n")
#define B_MIX R16("v_fma_mix_f32 v0, v8, -1.0, v12 op_sel_hi:[1,0,0]\n v_fma_mix_f32 v1, v9, -1.0, v12 op_sel_hi:[1,0,0]\n v_fma_mix_f32 v2, v10, -1.0, v12 op_sel_hi:[1,0,0]\n v_fma_mix_f32 v3, v11, -1.0, v12 op_sel_hi:[1,0,0]\n")
#define B_MIX_S R16("v_fma_mix_f32 v0, v8, -1.0, s4 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 v1, v9, -1.0, s5 op_sel_hi:[1,0,0]\n v_fma_mix_f32 v2, v10, -1.0, s6 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 v3, v11, -1.0, s7 op_sel_hi:[1,0,0]\n")
#define B_TRANSE_TAIL R16("v_sub_f32 v4, s4, v8\n v_add_f32 v0, v0, |v4|\n v_sub_f32 v5, s5, v9\n v_add_f32 v1, v1, |v5|\n")
#define B_MAX R16("v_max_f32 v0, v0, v8\n v_max_f32 v1, v1, v9\n v_max_f32 v2, v2, v10\n v_max_f32 v3, v3, v11\n")

#define B_SSNN R16("v_sub_f32 v4, s4, v8\n v_sub_f32 v5, s5, v9\n v_add_f32 v0, v0, |v4|\n v_add_f32 v1, v1, |v5|\n")
#define B_HEAD R16("v_add_f32 v4, s4, v8\n v_sub_f32 v4, v4, s5\n v_add_f32 v0, v0, |v4|\n v_add_f32 v1, v1, |v9|\n")
#define B_SAD_ADD R16("v_sad_u16 v0, v8, v9, v0\n v_add_f32 v4, v4, v10\n v_sad_u16 v1, v8, v11, v1\n v_add_f32 v5, v5, v10\n")
#define B_MAX_ADD R16("v_max_f32 v0, v0, v8\n v_add_f32 v4, v4, v10\n v_max_f32 v1, v1, v9\n v_add_f32 v5, v5, v10\n")
#define B_CVT_ADD R16("v_cvt_f32_f16 v0, v8\n v_add_f32 v4, v4, v10\n v_cvt_f32_f16 v1, v9\n v_add_f32 v5, v5, v10\n")
#define B_CMP R16("v_cmp_gt_f32 s[10:11], v0, v8\n v_cmp_gt_f32 s[12:13], v1, v8\n v_cmp_gt_f32 s[14:15], v2, v8\n v_cmp_gt_f32 s[16:17], v3, v8\n")
#define B_CMP_VCC R16("v_cmp_gt_f32 vcc, v0, v8\n v_cmp_gt_f32 vcc, v1, v8\n v_cmp_gt_f32 vcc, v2, v8\n v_cmp_gt_f32 vcc, v3, v8\n")
#define B_ADDCO R16("v_add_co_u32 v0, vcc, v8, v8\n v_add_co_u32 v1, vcc, v9, v9\n v_add_co_u32 v2, vcc, v10, v10\n v_add_co_u32 v3, vcc, v11, v11\n")
#define B_ADDC R16("v_addc_co_u32 v0, vcc, v0, v0, vcc\n v_addc_co_u32 v1, vcc, v1, v1, vcc\n v_addc_co_u32 v2, vcc, v2, v2, vcc\n v_addc_co_u32 v3, vcc, v3, v3, vcc\n")
#define B_LSHLADD R16("v_lshl_add_u32 v0, v8, 16, v9\n v_lshl_add_u32 v1, v9, 16, v9\n v_lshl_add_u32 v2, v10, 16, v9\n v_lshl_add_u32 v3, v11, 16, v9\n")
#define B_PERM R16("v_perm_b32 v0, v8, v9, v12\n v_perm_b32 v1, v9, v9, v12\n v_perm_b32 v2, v10, v9, v12\n v_perm_b32 v3, v11, v9, v12\n")
#define B_ALIGNBIT R16("v_alignbit_b32 v0, v0, v8, 31\n v_alignbit_b32 v1, v1, v9, 31\n v_alignbit_b32 v2, v2, v10, 31\n v_alignbit_b32 v3, v3, v11, 31\n")
#define B_MUL R16("v_mul_f32 v0, v0, v8\n v_mul_f32 v1, v1, v8\n v_mul_f32 v2, v2, v8\n v_mul_f32 v3, v3, v8\n")
#define B_PKFMA R16("v_pk_fma_f32 v[0:1], v[8:9], v[10:11], v[0:1]\n v_pk_fma_f32 v[2:3], v[8:9], v[10:11], v[2:3]\n v_pk_fma_f32 v[4:5], v[8:9], v[10:11], v[4:5]\n v_pk_fma_f32 v[6:7], v[8:9], v[10:11], v[6:7]\n")
#define B_MULU24 R16("v_mul_u32_u24 v0, v8, v9\n v_mul_u32_u24 v1, v9, v9\n v_mul_u32_u24 v2, v10, v9\n v_mul_u32_u24 v3, v11, v9\n")
#define B_OR R16("v_or_b32 v0, v0, v8\n v_or_b32 v1, v1, v9\n v_or_b32 v2, v2, v10\n v_or_b32 v3, v3, v11\n")
#define B_MOV R16("v_mov_b32 v0, v8\n v_mov_b32 v1, v9\n v_mov_b32 v2, v10\n v_mov_b32 v3, v11\n")
#define B_SUB_V R16("v_sub_f32 v0, v12, v8\n v_sub_f32 v1, v12, v9\n v_sub_f32 v2, v12, v10\n v_sub_f32 v3, v12, v11\n")
#define B_SUB_S1 R16("v_sub_f32 v0, s4, v8\n v_sub_f32 v1, s4, v9\n v_sub_f32 v2, s4, v10\n v_sub_f32 v3, s4, v11\n")

#define B_HYB R16("v_sad_u16 v0, s4, v9, v0\n v_sub_f32 v4, v13, v10\n v_sad_u16 v1, s5, v9, v1\n v_sad_u16 v0, s6, v11, v0\n v_add_f32 v5, v5, |v4|\n v_sad_u16 v1, s7, v12, v1\n")
#define B_HYB2 R16("v_sad_u16 v0, s4, v9, v0\n v_sub_f32 v4, v13, v10\n v_sad_u16 v1, s5, v9, v1\n v_add_f32 v5, v5, |v4|\n v_sad_u16 v0, s6, v11, v0\n v_sub_f32 v6, v13, v12\n v_sad_u16 v1, s7, v12, v1\n v_add_f32 v7, v7, |v6|\n")
#define B_SAD4S R16("v_sad_u16 v0, s4, v9, v0\n v_sad_u16 v1, s5, v9, v1\n v_sad_u16 v0, s6, v11, v0\n v_sad_u16 v1, s7, v12, v1\n")

#define KERNEL(name, body)                                                                          \
    __global__ __launch_bounds__(64) void name(int iters, unsigned long long* out) {                \
        unsigned long long t0 = __builtin_readcyclecounter();                                       \
        for (int i = 0; i < iters; ++i)                                                             \
            asm volatile(body ::: "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", \
                         "v11", "v12", "v13", "v14", "v15", "s4", "s5", "s6", "s7", "s10", "s11", "s12", "s13", "s14", "s15", "s16", "s17", "vcc");         \
        unsigned long long t1 = __builtin_readcyclecounter();                                       \
        if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;                                            \
    }

KERNEL(k_add, B_ADD)
KERNEL(k_add_abs, B_ADD_ABS)
KERNEL(k_sub_s, B_SUB_S)
KERNEL(k_fma, B_FMA)
KERNEL(k_fma_s, B_FMA_S)
KERNEL(k_cvt, B_CVT)
KERNEL(k_cvt_hi, B_CVT_HI)
KERNEL(k_lshl, B_LSHL)
KERNEL(k_and, B_AND)
KERNEL(k_addu, B_ADDU)
KERNEL(k_mix, B_MIX)
KERNEL(k_mix_s, B_MIX_S)
KERNEL(k_transe_tail, B_TRANSE_TAIL)
KERNEL(k_max, B_MAX)

KERNEL(k_ssnn, B_SSNN)
KERNEL(k_head, B_HEAD)
KERNEL(k_sad_add, B_SAD_ADD)
KERNEL(k_max_add, B_MAX_ADD)
KERNEL(k_cvt_add, B_CVT_ADD)
KERNEL(k_cmp, B_CMP)
KERNEL(k_cmp_vcc, B_CMP_VCC)
KERNEL(k_addco, B_ADDCO)
KERNEL(k_addc, B_ADDC)
KERNEL(k_lshladd, B_LSHLADD)
KERNEL(k_perm, B_PERM)
KERNEL(k_alignbit, B_ALIGNBIT)
KERNEL(k_mul, B_MUL)
KERNEL(k_pkfma, B_PKFMA)
KERNEL(k_mulu24, B_MULU24)
KERNEL(k_or, B_OR)
KERNEL(k_mov, B_MOV)
KERNEL(k_sub_v, B_SUB_V)
KERNEL(k_sub_s1, B_SUB_S1)

KERNEL(k_hyb, B_HYB)
KERNEL(k_hyb2, B_HYB2)
KERNEL(k_sad4s, B_SAD4S)

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
    printf("%-34s waves/SIMD=%d  %.3f ms  -> %.2f cyc/inst/SIMD @2.4GHz\n", name, waves_per_simd, ms,
           2.4e9 / (total / (ms * 1e-3) / 1024));
    CK(hipFree(d));
    return 0;
}

int main() {
    for (int w = 2; w <= 5; ++w) {
        run("add f32 (VOP2)", k_add, w);
        run("add f32 |abs| (VOP3)", k_add_abs, w);
        run("sub f32 sgpr - vgpr", k_sub_s, w);
        run("fma f32", k_fma, w);
        run("fma f32 sgpr operand", k_fma_s, w);
        run("max f32", k_max, w);
        run("cvt f32 <- f16 (lo)", k_cvt, w);
        run("cvt f32 <- f16 (hi, sdwa)", k_cvt_hi, w);
        run("lshlrev_b32 16", k_lshl, w);
        run("and_b32 literal", k_and, w);
        run("add_u32", k_addu, w);
        run("fma_mix f32 <- f16 x -1 + vgpr", k_mix, w);
        run("fma_mix f32 <- f16 x -1 + sgpr", k_mix_s, w);
        run("sub(sgpr) + add|abs| pairs", k_transe_tail, w);
        run("sub(s) sub(s) add|.| add|.|", k_ssnn, w);
        run("add(s) sub(s) add|.| add|.| (head)", k_head, w);
        run("sub f32 vgpr - vgpr, new dst", k_sub_v, w);
        run("sub f32 one sgpr - vgpr, new dst", k_sub_s1, w);
        run("mov b32", k_mov, w);
        run("or b32", k_or, w);
        run("mul f32", k_mul, w);
        run("pk_fma f32 (2 fma each)", k_pkfma, w);
        run("sad_u16 / add f32 alternating", k_sad_add, w);
        run("max f32 / add f32 alternating", k_max_add, w);
        run("cvt f16 / add f32 alternating", k_cvt_add, w);
        run("cmp_gt f32 -> sgpr pairs", k_cmp, w);
        run("cmp_gt f32 -> vcc", k_cmp_vcc, w);
        run("add_co u32 (carry out)", k_addco, w);
        run("addc_co u32 (carry in + out)", k_addc, w);
        run("lshl_add u32", k_lshladd, w);
        run("perm b32", k_perm, w);
        run("alignbit b32", k_alignbit, w);
        run("mul_u32_u24", k_mulu24, w);
        run("4 sad(sgpr), 2 chains (4 instr)", k_sad4s, w);
        run("4 sad(sgpr) + sub + add|.| (6 instr)", k_hyb, w);
        run("4 sad(sgpr) + 2 sub + 2 add|.| (8)", k_hyb2, w);
    }
    return 0;
}
