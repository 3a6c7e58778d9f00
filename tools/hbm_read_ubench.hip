// hbm_read.hip -- microbenchmark: what streaming-read rate does a wave-per-32KB-tile kernel reach on
// MI355X, as a function of access pattern, registers (occupancy) and loads in flight?
// Build: hipcc --offload-arch=gfx950 -O3 -o hbm_read hbm_read.hip ; run: ./hbm_read
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(err_), __LINE__); return 1; } } while (0)

// PATTERN 0: instruction i of a wave reads 1 KiB contiguous (2 rows of 512 B)
// PATTERN 1: instruction (s, i) reads 8 rows x 128 B (row stride 512 B)  [the rank kernel's pattern]
// KEEP: number of dummy VGPRs kept live to force a register count (occupancy)
template <int PATTERN, int NLOAD, int REGS>
__global__ __launch_bounds__(256) void read_kernel(const float4* __restrict__ src, float* __restrict__ out, long n_tiles, int iters_unused) {
    const int lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    float acc = 0.f;
    float pad[REGS + 1];
#pragma unroll
    for (int i = 0; i < REGS; ++i) pad[i] = (float)(lane + i);
    for (long tile = wave; tile < n_tiles; tile += (long)gridDim.x * 4) {
        const float4* base = src + tile * 2048;  // 32 KB = 2048 float4
        float4 v[NLOAD];
#pragma unroll
        for (int k = 0; k < NLOAD; ++k) {
            long idx;
            if (PATTERN == 0) idx = (long)k * 64 + lane;
            else { const int s = k / 8, i = k % 8; idx = (long)(8 * i + (lane >> 3)) * 32 + s * 8 + (lane & 7); }
            v[k] = base[idx];
        }
#pragma unroll
        for (int k = 0; k < NLOAD; ++k) acc += v[k].x + v[k].y + v[k].z + v[k].w;
#pragma unroll
        for (int i = 0; i < REGS; ++i) pad[i] = pad[i] * 1.0001f + acc;
    }
#pragma unroll
    for (int i = 0; i < REGS; ++i) acc += pad[i];
    if (acc == 123.456f) out[0] = acc;
}

template <int PATTERN, int NLOAD, int REGS>
int run(const char* name, const float4* d, float* out, long n_tiles, int blocks, hipStream_t st) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int w = 0; w < 2; ++w) read_kernel<PATTERN, NLOAD, REGS><<<blocks, 256, 0, st>>>(d, out, n_tiles, 0);
    CK(hipEventRecord(a, st));
    const int reps = 10;
    for (int r = 0; r < reps; ++r) read_kernel<PATTERN, NLOAD, REGS><<<blocks, 256, 0, st>>>(d, out, n_tiles, 0);
    CK(hipEventRecord(b, st));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    ms /= reps;
    const double bytes = (double)n_tiles * NLOAD * 1024;
    printf("%-44s blocks=%6d  %.3f ms  %.0f GB/s\n", name, blocks, ms, bytes / ms / 1e6);
    return 0;
}

int main() {
    const long n_tiles = 71875;  // 4.6 M rows x 512 B
    float4* d; float* out;
    CK(hipMalloc(&d, n_tiles * 32768)); CK(hipMalloc(&out, 4));
    CK(hipMemset(d, 1, n_tiles * 32768));
    hipStream_t st; CK(hipStreamCreate(&st));
    const int all = (int)((n_tiles + 3) / 4);
    run<0, 32, 1>("contig 32 loads, ~40 regs, 1 tile/wave", d, out, n_tiles, all, st);
    run<1, 32, 1>("rowstr 32 loads, ~40 regs, 1 tile/wave", d, out, n_tiles, all, st);
    run<0, 32, 100>("contig 32 loads, ~140 regs, 1 tile/wave", d, out, n_tiles, all, st);
    run<1, 32, 100>("rowstr 32 loads, ~140 regs, 1 tile/wave", d, out, n_tiles, all, st);
    run<1, 32, 100>("rowstr 32 loads, ~140 regs, grid 768", d, out, n_tiles, 768, st);
    run<1, 32, 100>("rowstr 32 loads, ~140 regs, grid 1536", d, out, n_tiles, 1536, st);
    run<1, 32, 100>("rowstr 32 loads, ~140 regs, grid 3072", d, out, n_tiles, 3072, st);
    run<0, 32, 100>("contig 32 loads, ~140 regs, grid 768", d, out, n_tiles, 768, st);
    run<0, 32, 100>("contig 32 loads, ~140 regs, grid 1536", d, out, n_tiles, 1536, st);
    run<0, 8, 100>("contig 8 loads(8KB/tile), ~140 regs, all", d, out, n_tiles, all, st);
    run<0, 16, 100>("contig 16 loads, ~140 regs, all", d, out, n_tiles, all, st);
    run<0, 32, 200>("contig 32 loads, ~240 regs, all", d, out, n_tiles, all, st);
    run<0, 32, 200>("contig 32 loads, ~240 regs, grid 512", d, out, n_tiles, 512, st);
    return 0;
}
